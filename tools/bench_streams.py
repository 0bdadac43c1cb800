#!/usr/bin/env python3
"""Experiment: does running consecutive batches on 2-3 HIP streams (one context each) overlap the HBM-bound ScaleDown
chain of one batch with the VALU-bound detection/description of another?  Prints ms per 64x1080p batch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import capi, synth  # noqa: E402


def main():
    n, w, h = 64, 1920, 1080
    p = capi.ialign_up(w, 128)
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
    one = np.zeros((h, p), dtype=np.float32)
    one[:, :w] = synth.tile(1000, w, h, 1.0)
    for n_streams in (1, 2, 3):
        ctxs = [capi.Context(0) for _ in range(n_streams)]
        bufs = []
        for c in ctxs:
            img = capi.DeviceBuffer(c, n * h * p * 4)
            for i in range(n):
                c.h2d(img.ptr + i * h * p * 4, one)
            pts = capi.DeviceBuffer(c, n * prm.max_pts * 588)
            cnt = capi.DeviceBuffer(c, 4 * n)
            bufs.append((img, pts, cnt))
        def step(k):
            c = ctxs[k % n_streams]
            img, pts, cnt = bufs[k % n_streams]
            c.extract_batch(img.ptr, n, w, h, p, h * p, prm, pts.ptr, cnt.ptr)
        for k in range(6):
            step(k)
        for c in ctxs:
            c.synchronize()
        steps = 30
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        for c in ctxs:
            c.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print("%d stream(s): %.4f ms per batch  (%.1f Gpix/s)" % (n_streams, ms, n * w * h / ms / 1e6), flush=True)
        for b in bufs:
            for x in b:
                x.free()
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
