#!/usr/bin/env python3
"""cusift_pipe_* host to host, 64 x 1080p 8-bit frames per batch, by the number of batches in flight: python
tools/probe_pipe_depth.py [depth ...] (a fresh process per depth is the fair comparison: HIP maps streams onto its four
hardware queues in order of creation, so a second pipeline in one process sits on other queues than the first)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from cusift_amd import capi, synth
B, w, h = 64, 1920, 1080
prm_kw = dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
frames = torch.empty((B, h, w), dtype=torch.uint8).pin_memory()
one = np.clip(np.rint(synth.tile(1000, w, h, 1.0)), 0, 255).astype(np.uint8)
for i in range(B):
    frames[i] = torch.from_numpy(np.clip(np.rint(synth.tile(1000 + i, w, h, 1.0)), 0, 255).astype(np.uint8))
fr = frames.numpy()
for depth in [int(a) for a in sys.argv[1:]] or [4]:
    p = capi.Pipe(0, B, w, h, capi.default_params(**prm_kw), capi.PIPE_U8, depth=depth, records_capacity=300000)
    def run(steps):
        got = 0
        for _ in range(steps):
            if p.in_flight() == depth:
                got += len(p.collect()[0])
            p.submit(fr)
        while p.in_flight():
            got += len(p.collect()[0])
        return got
    run(depth + 2)
    t = time.perf_counter(); n = 30; g = run(n); dt = time.perf_counter() - t
    print("depth %d: %.4f ms per step, %.1f Gpix/s, %.1f M kp/s" % (depth, dt / n * 1e3, B * w * h / (dt / n) / 1e9, g / dt / 1e6), flush=True)
    p.close()
