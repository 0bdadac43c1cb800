#!/usr/bin/env python3
"""One extraction on ONE stream (a lone caller: cusift_params.concurrent_batches = 1) with and without the side stream
for octave 0's detection (cusift_ctx_set_policy(CUSIFT_POLICY_SIDE_STREAM, 0 / 1)), interleaved on one box: ms per cusift_extract_batch of
B x 1080p (5 octaves, the benchmark's images), back to back, and the latency of one call.

    python tools/probe_octave_overlap.py [reps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import capi, synth  # noqa: E402


def make_ctx(mode):
    c = capi.Context(0)
    c.set_policy(capi.POLICY_SIDE_STREAM, mode)
    return c


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    w, h = 1920, 1080
    p = capi.ialign_up(w, 128)
    base = [np.pad(synth.tile(1000 + i, w, h, 1.0), ((0, 0), (0, p - w))) for i in range(4)]
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
    ctxs = {"one stream": make_ctx(0), "side stream": make_ctx(1)}
    for B in [int(x) for x in os.environ.get("PROBE_BATCHES", "1,4,16,64").split(",")]:
        imgs = np.stack([base[i % 4] for i in range(B)])
        state = {}
        for name, c in ctxs.items():
            state[name] = (capi.DeviceBuffer.from_numpy(c, imgs), capi.DeviceBuffer(c, B * prm.max_pts * 588),
                           capi.DeviceBuffer(c, 4 * B))

        def run(name, n, sync_each):
            c = ctxs[name]
            d, pts, cnt = state[name]
            c.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                c.extract_batch(d.ptr, B, w, h, p, h * p, prm, pts.ptr, cnt.ptr)
                if sync_each:
                    c.synchronize()
            c.synchronize()
            return (time.perf_counter() - t) / n * 1e3

        res = {k: ([], []) for k in ctxs}
        for name in ctxs:
            run(name, 5, False)
        for _ in range(5):  # interleaved rounds
            for name in ctxs:
                res[name][0].append(run(name, reps, False))
                res[name][1].append(run(name, reps, True))
        counts = {}
        for name, c in ctxs.items():
            counts[name] = int(np.minimum(state[name][2].to_numpy(np.uint32, (B,)), prm.max_pts).sum())
        assert len(set(counts.values())) == 1, counts
        for name in ctxs:
            print("B=%-3d %-11s back to back %.4f ms (min %.4f)   call + wait %.4f ms (min %.4f)   keypoints %d" % (
                B, name, float(np.median(res[name][0])), min(res[name][0]), float(np.median(res[name][1])),
                min(res[name][1]), counts[name]), flush=True)
        for name in ctxs:
            for b in state[name]:
                b.free()
    for c in ctxs.values():
        c.close()


if __name__ == "__main__":
    main()
