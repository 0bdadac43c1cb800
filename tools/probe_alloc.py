#!/usr/bin/env python3
"""Experiment: does the blur+DoG kernel's HBM rate depend on WHERE its buffers were allocated?
Re-allocates the DoG block several times in one process and times cusift_laplace_multi on each allocation."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import capi  # noqa: E402


def main():
    n, w, h = 64, 1920, 1080
    p = capi.ialign_up(w, 128)
    lib = capi.lib()
    with capi.Context(0) as ctx:
        img = capi.DeviceBuffer(ctx, n * h * p * 4)
        ctx.memset(img.ptr, 0, img.nbytes)
        dog_bytes = n * 7 * h * p * 4
        pad = int(os.environ.get("PROBE_PAD", "0"))
        keep = []
        for trial in range(int(os.environ.get("PROBE_TRIALS", "8"))):
            dog = capi.DeviceBuffer(ctx, dog_bytes + pad)
            if trial % 2 == 1:  # perturb the allocator: hold a block of odd size between trials
                keep.append(capi.DeviceBuffer(ctx, (37 + 11 * trial) << 20))
            for _ in range(2):
                capi.check(lib.cusift_laplace_multi(ctx.handle, img.ptr, w, h, p, h * p, 1.0, dog.ptr, 7 * h * p, n))
            ctx.synchronize()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                capi.check(lib.cusift_laplace_multi(ctx.handle, img.ptr, w, h, p, h * p, 1.0, dog.ptr, 7 * h * p, n))
            ctx.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            print("trial %d  dog@0x%x  %.4f ms  %.0f GB/s" % (trial, dog.ptr, ms, n * w * h * 32 / ms / 1e6), flush=True)
            dog.free()


if __name__ == "__main__":
    main()
