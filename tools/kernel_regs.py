#!/usr/bin/env python3
"""Registers, spills, scratch, LDS and waves per SIMD of every kernel in one .hip unit, from the metadata hipcc emits
for the product's flags (runs here: hipcc cross-compiles without a GPU).

    python tools/kernel_regs.py sift_stencils.hip [needle ...]      # rows whose name contains any needle
    python tools/kernel_regs.py sift_stencils.hip --asm /tmp/x.s    # also keeps the assembly

gfx950: 512 unified registers per SIMD lane => waves per SIMD = min(8, 512 // align(vgpr + agpr, 8)).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import build as B  # noqa: E402


def assembly(src, extra=()):
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [B.find_hipcc()] + flags + list(extra) + ["-S", "--cuda-device-only", "-o", "-", os.path.join(B.CSRC, src)]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def kernels(asm):
    """[(demangled-ish name, dict)] from the amdhsa.kernels metadata block."""
    out = []
    for block in re.split(r"\n  - \.agpr_count:", asm)[1:]:
        block = ".agpr_count:" + block
        d = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                    "private_segment_fixed_size", "group_segment_fixed_size"):
            m = re.search(r"\.%s:\s+(\d+)" % key, block)
            d[key] = int(m.group(1)) if m else -1
        m = re.search(r"\.name:\s+(\S+)", block)
        d["name"] = m.group(1) if m else "?"
        out.append(d)
    return out


def waves_per_simd(d):
    regs = (d["vgpr_count"] + max(0, d["agpr_count"]) + 7) // 8 * 8
    return min(8, 512 // max(8, regs))


def main():
    args = [a for a in sys.argv[1:]]
    keep = None
    if "--asm" in args:
        i = args.index("--asm")
        keep = args[i + 1]
        del args[i:i + 2]
    src, needles = args[0], args[1:]
    asm = assembly(src)
    if keep:
        with open(keep, "w") as f:
            f.write(asm)
    print("%-78s %5s %5s %5s %6s %6s %8s %7s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds", "w/SIMD"))
    for d in kernels(asm):
        if needles and not any(n in d["name"] for n in needles):
            continue
        print("%-78s %5d %5d %5d %6d %6d %8d %7d %5d" % (d["name"][:78], d["vgpr_count"], d["agpr_count"], d["sgpr_count"],
                                                       d["vgpr_spill_count"], d["sgpr_spill_count"],
                                                       d["private_segment_fixed_size"], d["group_segment_fixed_size"],
                                                       waves_per_simd(d)))


if __name__ == "__main__":
    main()
