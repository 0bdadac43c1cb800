#!/usr/bin/env python3
"""bench.py's own sequence -- pre-flight, W warm-up steps, fence, K timed steps, then repeats -- with an event behind
every step: where inside a region does the time go?  Prints, per region, the wall ms per step and the per-stream
completion spacing (ms per step) of its steps in order.

    python tools/probe_bench_regions.py [preflight=24] [warmup=5] [steps=20]
"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cusift_amd import synth  # noqa: E402
from cusift_amd.batch import PipelinedExtractor  # noqa: E402


def main():
    pre = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    B, w, h, E = 64, 1920, 1080, 4
    pipe = PipelinedExtractor(B, w, h, n_streams=E, num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
    imgs = np.stack([synth.tile(1000 + i, w, h, 1.0) for i in range(8)] * 8)
    d = pipe.images_from_numpy(imgs)
    torch.cuda.synchronize()

    def region(n, label, timing_events=True):
        evs = []
        t0 = time.perf_counter()
        for _ in range(n):
            k = pipe.submitted
            st = pipe.streams[k % E]
            pipe.submit(d)
            if timing_events:
                e = torch.cuda.Event(enable_timing=True)
                e.record(st)
                evs.append(e)
        pipe.synchronize()
        wall = (time.perf_counter() - t0) / max(1, n) * 1e3
        per = [evs[i - E].elapsed_time(evs[i]) / E for i in range(E, len(evs))] if timing_events else []
        print("%-28s %2d steps: wall %.4f ms/step | %s" % (label, n, wall, " ".join("%.2f" % x for x in per)), flush=True)

    region(pre, "pre-flight")
    region(warm, "warm-up")
    torch.cuda.synchronize()
    region(K, "timed")
    for i in range(4):
        region(K, "repeat %d" % i)
    time.sleep(0.05)
    region(K, "after 50 ms idle")
    region(K, "right behind it")
    region(K, "no per-step events", timing_events=False)
    region(K, "no per-step events", timing_events=False)


if __name__ == "__main__":
    main()
