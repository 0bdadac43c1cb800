"""Builds libcusift_amd.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

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
SOURCES = ["sift_context.hip", "sift_stages.hip", "sift_driver.hip", "sift_stencils.hip", "sift_keypoints.hip", "sift_match.hip",
           "sift_frontend.hip", "sift_homography.hip", "sift_comm.hip", "sift_tiled.hip", "sift_pipe.hip"]
HEADERS = [os.path.join(CSRC, "detect_chunk.inc"), os.path.join(CSRC, "sift_types.h"), os.path.join(CSRC, "sift_device.h"), os.path.join(CSRC, "sift_math.h"), os.path.join(CSRC, "sift_internal.h"), os.path.join(CSRC, "sift_host.h"),
           os.path.join(HERE, "..", "include", "cusift_amd.h")]

# -ffp-contract=off: the only fused multiply-adds are the explicit fmaf() calls (see sift_types.h).
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


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


LAB_LIB = os.path.join(HERE, "libcusift_amd_lab.so")
STAMPS_LIB = os.path.join(HERE, "libcusift_amd_stamps.so")  # --stamps: -DCUSIFT_STAMPS, for tools/describe_stamps.py


def build(force=False, verbose=False, lab=False, stamps=False):
    """Compile the HIP extension for gfx950 if missing or older than its sources. Returns the path."""
    if stamps:
        cmd = [find_hipcc()] + HIPCC_FLAGS + ["-DCUSIFT_STAMPS", "-o", STAMPS_LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return STAMPS_LIB
    if lab:
        cmd = [find_hipcc()] + HIPCC_FLAGS + ["-DCUSIFT_LAB", "-o", LAB_LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return LAB_LIB
    if not force and not is_stale():
        return LIB
    # build into a temporary file and rename it onto LIB: a concurrent rank that loads the library (or builds it too)
    # never sees a half-written shared object
    tmp = "%s.%d.tmp" % (LIB, os.getpid())
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv, stamps="--stamps" in sys.argv))
