#!/usr/bin/env python3
"""How long does ONE rank of the strip-tiled extraction take on a GPU of its own?  8192 x 8192 over 8 virtual ranks on one
device: the bands are built once (ScaleDown + virtual halo exchange), then rank 3's steps are timed alone -- the
per-octave ScaleDown of its band (no exchange: its neighbours' rows are already there) and cusift_tiled_process
(detection + description of its bands).

    python tools/probe_tiled_rank.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cusift_amd import capi, synth  # noqa: E402
from cusift_amd.tiling import StripExtractor  # noqa: E402


def main():
    W = H = 8192
    P = 8
    dev = torch.device("cuda", 0)
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=1 << 19)
    img = synth.tile(4242, W, H, preblur=1.0)
    exts = [StripExtractor(k, P, W, H, prm, device=dev) for k in range(P)]
    b = exts[0].plan.bounds
    strips = [torch.from_numpy(img[b[k]:b[k + 1]]).to(dev) for k in range(P)]
    pl = exts[0].plan
    tiles = [e.tiled for e in exts]
    for e, s in zip(exts, strips):
        e.load_strip(s)
    for o in range(min(pl.collapse + 1, pl.n_oct)):
        if o > 0:
            for e in exts:
                e.tiled.build_octave(o)
        capi.tiled_exchange_virtual(tiles, o)
    torch.cuda.synchronize()
    e = exts[3]

    def timed(fn, reps=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    def build():
        for o in range(1, min(pl.collapse + 1, pl.n_oct)):
            e.tiled.build_octave(o)

    t_build = timed(build)
    t_proc = timed(e.process)
    n = len(e.result())
    print("rank 3 of 8, 8192^2: ScaleDown of its bands %.4f ms, cusift_tiled_process %.4f ms (%d keypoints); "
          "collapse octave %d of %d" % (t_build, t_proc, n, pl.collapse, pl.n_oct))


if __name__ == "__main__":
    main()
