"""Builds libcusift_amd.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc -- by running the top-level Makefile, the
one recipe of this repository (a C++ caller needs no Python: `make`, or the CMakeLists.txt next to it).

    python -m cusift_amd.build [--force] [--lab] [--stamps]

--lab builds libcusift_amd_lab.so with -DCUSIFT_LAB: the same kernels, plus the tuning overrides the A/B scripts under
tools/ read from the environment (CUSIFT_*_ROWS_*, CUSIFT_DETECT_WAVES, ...; select it with CUSIFT_AMD_LIB=<path>).  The
product library reads none of them.

--stamps builds libcusift_amd_stamps.so with -DCUSIFT_STAMPS: describe_all_kernel with shader-clock stamps at its phase
boundaries (tools/describe_stamps.py).

The shared object lands next to this file so that it travels to the GPU box with the source tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcusift_amd.so")

# The flags of the Makefile (HIPFLAGS there), for the tools that compile a translation unit to assembly themselves
# (tools/isa_mix.py, tools/kernel_regs.py).  -ffp-contract=off: the only fused multiply-adds are the explicit fmaf() calls (see sift_types.h).
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    # no automatic packing of adjacent scalar fp32 operations into v_pk_*_f32: measured, the keypoint kernel is 2.7 %
    # faster without it (a packed operation costs 4.2-4.4 cycles per SIMD against 2 x 2.65 for two plain ones without an
    # SGPR operand -- tools/microbench/valu_rate.hip -- and the compiler's packing adds v_pk_mov / v_mov_b64 shuffles).  The blur kernels pack BY HAND (scale pairs, one SGPR tap pair
    # per operation) because there the halved instruction count wins (measured both ways).
    "-fno-slp-vectorize",
    "-fPIC",
    "-shared",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


ROOT = os.path.dirname(HERE)
LAB_LIB = os.path.join(HERE, "libcusift_amd_lab.so")
STAMPS_LIB = os.path.join(HERE, "libcusift_amd_stamps.so")  # --stamps: -DCUSIFT_STAMPS, for tools/describe_stamps.py


def is_stale():
    """make's own judgement (`make -q`): objects older than their sources or headers, or no library yet."""
    return subprocess.call(["make", "-q", "-C", ROOT, "all", "HIPCC=" + find_hipcc()], stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL) != 0


def build(force=False, verbose=False, lab=False, stamps=False):
    """Compile the HIP extension for gfx950 if missing or older than its sources (make decides). Returns the path.
    One object per translation unit under build/obj, compiled in parallel; the link goes to a temporary file that is
    renamed onto the library, so a process that LOADS the library never sees a half-written shared object.  Builders are
    serialised: the objects and the link's temporary file have fixed names, so two ranks running this at once would
    otherwise link each other's half-written objects -- the whole make runs under an exclusive flock on build/.lock (the
    second builder then finds everything up to date)."""
    import fcntl

    target, lib = ("stamps", STAMPS_LIB) if stamps else (("lab", LAB_LIB) if lab else ("all", LIB))
    cmd = ["make", "-C", ROOT, "-j%d" % max(1, min(8, os.cpu_count() or 1)), target, "HIPCC=" + find_hipcc()]
    if force:
        cmd.insert(1, "-B")
    if verbose:
        print(" ".join(cmd))
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    with open(os.path.join(ROOT, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv, stamps="--stamps" in sys.argv))
