#!/usr/bin/env python3
"""How long after a cold start does a pipelined 64 x 1080p step reach its steady rate?  Four streams, one event per
step on its stream; prints the time between consecutive steps' completions on each stream (= 4 steps of device work)
for the first 120 steps of the process, then again after an idle second.  (bench.py's first timed region -- 5 warm-up
steps, 20 timed -- sits 6 % above its repeats: this shows where those 6 % are.)

    python tools/probe_rampup.py
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import capi, synth  # noqa: E402


def main():
    n, w, h, E = 64, 1920, 1080, 4
    p = capi.ialign_up(w, 128)
    rows = []
    for i in range(8):
        r = np.zeros((h, p), dtype=np.float32)
        r[:, :w] = synth.tile(1000 + i, w, h, 1.0)
        rows.append(r)
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768, concurrent_batches=E)
    lib = capi.lib()
    ctxs = [capi.Context(0) for _ in range(E)]
    bufs = []
    for c in ctxs:
        img = capi.DeviceBuffer(c, n * h * p * 4)
        for i in range(n):
            c.h2d(img.ptr + i * h * p * 4, rows[i % 8])
        bufs.append((img, capi.DeviceBuffer(c, n * prm.max_pts * 588), capi.DeviceBuffer(c, 4 * n)))
        c.synchronize()

    def run(steps, label, stagger_ms=0.0):
        evs = []
        for k in range(steps):
            c = ctxs[k % E]
            img, pts, cnt = bufs[k % E]
            if stagger_ms and 0 < k < E:  # the first step of streams 1 .. E-1 starts a quarter of a step after its neighbour's
                t = time.perf_counter()
                while (time.perf_counter() - t) * 1e3 < stagger_ms:
                    pass
            c.extract_batch(img.ptr, n, w, h, p, h * p, prm, pts.ptr, cnt.ptr)
            ev = C.c_void_p()
            capi.check(lib.cusift_event_create(c.handle, C.byref(ev)))
            capi.check(lib.cusift_event_record(ev, c.handle))
            evs.append(ev)
        for c in ctxs:
            c.synchronize()
        # per stream: ms between the completions of its consecutive steps = E steps of device work
        per = []
        for k in range(E, steps):
            ms = C.c_float(0)
            capi.check(lib.cusift_event_elapsed_ms(evs[k - E], evs[k], C.byref(ms)))
            per.append(ms.value / E)
        for ev in evs:
            lib.cusift_event_destroy(ev)
        chunks = [per[i:i + 10] for i in range(0, len(per), 10)]
        print(label, " ".join("%.3f" % (sum(c) / len(c)) for c in chunks if c), "(ms per step, means of 10 consecutive steps)", flush=True)

    run(120, "cold start :")
    run(120, "right after:")
    time.sleep(1.0)
    run(120, "after 1 s idle:")
    time.sleep(0.05)
    run(60, "after 50 ms idle:")
    time.sleep(0.05)
    run(60, "after 50 ms idle, streams started 0.25 ms apart:", stagger_ms=0.25)
    time.sleep(0.05)
    run(60, "after 50 ms idle, streams started 0.5 ms apart:", stagger_ms=0.5)
    time.sleep(0.05)
    run(60, "after 50 ms idle:")
    for c in ctxs[:1]:  # keep the chip busy for 30 ms with ONE stream, then all four with no gap
        for _ in range(25):
            img, pts, cnt = bufs[0]
            c.extract_batch(img.ptr, n, w, h, p, h * p, prm, pts.ptr, cnt.ptr)
    run(60, "behind 25 single-stream steps, no gap:")


if __name__ == "__main__":
    main()
