#!/usr/bin/env python3
"""Where a wave of describe_all_kernel spends its time: shader-clock stamps at the phase boundaries of every keypoint.

    python -m cusift_amd.build --stamps          # here: libcusift_amd_stamps.so (-DCUSIFT_STAMPS; never the product)
    python tools/describe_stamps.py [content]    # on the GPU; content: tile (default), blobs, raw

Prints cycles per keypoint per wave by phase (lane 0's s_memtime deltas, summed over all waves of all launches and
divided by the keypoints described).  The stamps cost ~10 instructions each, ~8 % of the launch."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CUSIFT_AMD_LIB", os.path.join(ROOT, "cusift_amd", "libcusift_amd_stamps.so"))
from cusift_amd import capi, synth  # noqa: E402

PHASES = [
    (12, "loop back edge (+ prologue, once per wave)"),
    (0, "head: record fields, geometry, patch loads issued"),
    (1, "orientation prep (Gaussian table, lattice weights) beside the loads"),
    (2, "wait for the patch (vmcnt 0)"),
    (3, "next item: cursor value, its image, load of the segment ends issued"),
    (4, "orientation: taps, atan2, sqrt, bin, weight"),
    (5, "orientation: list posted + walked"),
    (6, "orientation: smoothing, peak, interpolation"),
    (14, "between the stages: next item's segment, load of its head issued"),
    (7, "descriptor phase 1: sincos, 16 taps, atan2, sqrt per lane"),
    (8, "descriptor: vertical pass"),
    (9, "descriptor: horizontal pass"),
    (10, "descriptor: normalisation (two tree sums)"),
    (11, "record stores"),
]


def main():
    content = sys.argv[1] if len(sys.argv) > 1 else "tile"
    n, w, h = 64, 1920, 1080
    p = capi.ialign_up(w, 128)
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
    ctx = capi.Context(0)
    img = capi.DeviceBuffer(ctx, n * h * p * 4)
    one = np.zeros((h, p), dtype=np.float32)
    for i in range(n):
        if content == "blobs":
            one[:, :w] = synth.blobs(1000 + i, w, h)
        elif content == "raw":
            one[:, :w] = synth.tile(1000 + i, w, h, 0.0)
        else:
            one[:, :w] = synth.tile(1000 + i, w, h, 1.0)
        ctx.h2d(img.ptr + i * h * p * 4, one)
    pts = capi.DeviceBuffer(ctx, n * prm.max_pts * 588)
    cnt = capi.DeviceBuffer(ctx, 4 * n)
    lib = capi.lib()
    lib.cusift_stamps_read.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.cusift_stamps_read.restype = C.c_int
    out = (C.c_ulonglong * 16)()
    for _ in range(3):
        ctx.extract_batch(img.ptr, n, w, h, p, h * p, prm, pts.ptr, cnt.ptr)
    ctx.synchronize()
    assert lib.cusift_stamps_read(out) == 0  # reads and clears
    steps = 10
    for _ in range(steps):
        ctx.extract_batch(img.ptr, n, w, h, p, h * p, prm, pts.ptr, cnt.ptr)
    ctx.synchronize()
    assert lib.cusift_stamps_read(out) == 0
    kp = out[13]
    total = sum(out[i] for i, _ in PHASES)
    print("content %s: %d keypoints per step, %.0f cycles per keypoint per wave" % (content, kp // steps, total / kp))
    for i, name in PHASES:
        print("  %8.0f  %5.1f %%  %s" % (out[i] / kp, 100.0 * out[i] / total, name))


if __name__ == "__main__":
    main()
