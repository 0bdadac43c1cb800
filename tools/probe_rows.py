#!/usr/bin/env python3
"""Experiment: rows-per-wave (chunk height) of the strip kernels vs time, per stage, in one process (interleaved
trials so that slow drifts of the device hit every setting alike).  PROBE_STAGE = laplace | detect | find | down."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import capi, synth  # noqa: E402


def main():
    stage = os.environ.get("PROBE_STAGE", "laplace")
    n = int(os.environ.get("PROBE_N", "64"))
    w, h = int(os.environ.get("PROBE_W", "1920")), int(os.environ.get("PROBE_H", "1080"))
    rows_list = [int(x) for x in os.environ.get("PROBE_ROWS", "2,3,4,5,6,8,10,12,16,24,32").split(",")]
    p = capi.ialign_up(w, 128)
    lib = capi.lib()
    env = {"laplace": "LAPLACE", "detect": "DETECT", "find": "FINDPOINTS", "down": "SCALEDOWN"}[stage]
    with capi.Context(0) as ctx:
        one = np.zeros((h, p), dtype=np.float32)
        one[:, :w] = synth.tile(1000, w, h, 1.0)
        img = capi.DeviceBuffer(ctx, n * h * p * 4)
        for i in range(n):
            ctx.h2d(img.ptr + i * h * p * 4, one)
        dog = capi.DeviceBuffer(ctx, n * 7 * h * p * 4) if stage in ("laplace", "find") else None
        pts = capi.DeviceBuffer(ctx, n * 32768 * 588)
        cnt = capi.DeviceBuffer(ctx, 4 * n)
        half = capi.DeviceBuffer(ctx, n * (h // 2) * capi.ialign_up(w // 2, 128) * 4)
        if stage == "find":
            capi.check(lib.cusift_laplace_multi(ctx.handle, img.ptr, w, h, p, h * p, 1.0, dog.ptr, 7 * h * p, n))

        def launch(ctx):
            if stage == "laplace":
                capi.check(lib.cusift_laplace_multi(ctx.handle, img.ptr, w, h, p, h * p, 1.0, dog.ptr, 7 * h * p, n))
            elif stage == "detect":
                ctx.memset(cnt.ptr, 0, 4 * n)
                capi.check(lib.cusift_detect_multi(ctx.handle, img.ptr, w, h, p, h * p, 1.0, 3.0, 10.0, 1.0, pts.ptr,
                                                   32768, cnt.ptr, n))
            elif stage == "find":
                ctx.memset(cnt.ptr, 0, 4 * n)
                capi.check(lib.cusift_find_points_multi(ctx.handle, dog.ptr, w, h, p, 7 * h * p, 3.0, 10.0, 1.0,
                                                        pts.ptr, 32768, cnt.ptr, n))
            else:
                hp = capi.ialign_up(w // 2, 128)
                capi.check(lib.cusift_scale_down(ctx.handle, half.ptr, hp, (h // 2) * hp, img.ptr, w, h, p, h * p, n,
                                                 0.5))

        best = {r: [] for r in rows_list}
        for rep in range(int(os.environ.get("PROBE_REPS", "4"))):
            for r in rows_list:
                os.environ["CUSIFT_%s_ROWS_LO" % env] = str(r)
                os.environ["CUSIFT_%s_ROWS_HI" % env] = str(r)
                with capi.Context(0) as tuned:  # the tuning knobs are read once, when a context is created
                    launch(tuned)
                    tuned.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(8):
                        launch(tuned)
                    tuned.synchronize()
                    best[r].append((time.perf_counter() - t0) / 8 * 1e3)
        for r in rows_list:
            v = np.array(best[r])
            print("%s %dx%dx%d rows=%-3d  median %.4f ms  min %.4f  max %.4f" % (stage, n, w, h, r, np.median(v), v.min(),
                                                                           v.max()), flush=True)


if __name__ == "__main__":
    main()
