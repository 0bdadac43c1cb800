"""Times cusift_match on n x n synthetic unit descriptors: tools/bench_match.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from cusift_amd import capi
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer

ctx = capi.Context(0)
rng = np.random.default_rng(0)
sizes = [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 32768]
for n in sizes:
    p = np.zeros(n, dtype=SIFT_POINT_DTYPE)
    d = np.abs(rng.normal(size=(n, 128))).astype(np.float32)
    p["data"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    d1 = DeviceBuffer.from_numpy(ctx, p)
    d2 = DeviceBuffer.from_numpy(ctx, p[::-1].copy())
    for _ in range(3):
        ctx.match(d1.ptr, n, d2.ptr, n, 1)
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        ctx.match(d1.ptr, n, d2.ptr, n, 1)
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 10
    ok = bool((d1.to_numpy(SIFT_POINT_DTYPE, n)["match"] == np.arange(n)[::-1]).all())
    print("n=%d: %.3f ms, %.1f TFLOP/s (2*n*n*128 flop), %.1f G pairs/s, self-match %s"
          % (n, dt * 1e3, 2 * n * n * 128 / dt / 1e12, n * n / dt / 1e9, "ok" if ok else "WRONG"))
    d1.free()
    d2.free()
