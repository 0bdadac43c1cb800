#!/usr/bin/env python3
"""Same-box, interleaved A/B of CUSIFT_POLICY_PYRAMID_IN_DETECT (the next octave's image as a by-product of the
detection): ms per 64 x 1080p batch for policy 0 (ScaleDown chain first), 1 (octave 0's detection writes octave 1) and
2 (every detection writes the next octave), for a lone caller (one stream) and a pipelining caller (four streams,
concurrent_batches = 4: the benchmark's timed region).

    python tools/ab_pyramid.py [reps=5] [content=tile|blobs|raw] [n=64]
    AB_POLICIES=2 AB_STREAMS=1 rocprofv3 --kernel-trace --stats ... -- python3 tools/ab_pyramid.py 1   # one variant's kernels
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cusift_amd import capi, synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    content = sys.argv[2] if len(sys.argv) > 2 else "tile"
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    w, h = 1920, 1080
    p = capi.ialign_up(w, 128)
    n_distinct = min(n, 8)
    if content == "tile":
        imgs = [synth.tile(1000 + i, w, h, 1.0) for i in range(n_distinct)]
    elif content == "raw":
        imgs = [synth.tile(1000 + i, w, h, 0.0) for i in range(n_distinct)]
    else:
        imgs = [synth.blobs(1000 + i, w, h) for i in range(n_distinct)]
    rows = [np.zeros((h, p), dtype=np.float32) for _ in imgs]
    for r, i in zip(rows, imgs):
        r[:, :w] = i
    policies = tuple(int(x) for x in os.environ.get("AB_POLICIES", "0,1,2").split(","))  # (a kernel trace wants one)
    results = {}
    for n_streams in tuple(int(x) for x in os.environ.get("AB_STREAMS", "1,4").split(",")):
        prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768,
                                  concurrent_batches=n_streams if n_streams > 1 else 1)
        # one set of contexts per policy, all alive at once so that the trials interleave on the same device state
        sets = {}
        for pol in policies:
            ctxs = [capi.Context(0) for _ in range(n_streams)]
            bufs = []
            for c in ctxs:
                c.set_policy(capi.POLICY_PYRAMID_IN_DETECT, pol)
                img = capi.DeviceBuffer(c, n * h * p * 4)
                for i in range(n):
                    c.h2d(img.ptr + i * h * p * 4, rows[i % n_distinct])
                pts = capi.DeviceBuffer(c, n * prm.max_pts * 588)
                cnt = capi.DeviceBuffer(c, 4 * n)
                bufs.append((img, pts, cnt))
            sets[pol] = (ctxs, bufs)

        def run(pol, steps):
            ctxs, bufs = sets[pol]
            for k in range(steps):
                c = ctxs[k % n_streams]
                img, pts, cnt = bufs[k % n_streams]
                c.extract_batch(img.ptr, n, w, h, p, h * p, prm, pts.ptr, cnt.ptr)
            for c in ctxs:
                c.synchronize()

        counts = {}
        for pol in policies:
            run(pol, 2 * n_streams)
            ctxs, bufs = sets[pol]
            counts[pol] = bufs[0][2].to_numpy(np.uint32, (n,)).copy()
        for pol in policies[1:]:
            assert np.array_equal(counts[pol], counts[policies[0]]), "policy %d: other keypoint counts" % pol
        steps = 40
        for rep in range(reps):
            for pol in policies:
                t0 = time.perf_counter()
                run(pol, steps)
                ms = (time.perf_counter() - t0) / steps * 1e3
                results.setdefault((n_streams, pol), []).append(ms)
        for pol in policies:
            v = sorted(results[(n_streams, pol)])
            print("%d stream(s) policy %d (%s, %d x 1080p, %d keypoints/step): median %.4f ms  min %.4f  max %.4f  (%.1f Gpix/s)"
                  % (n_streams, pol, content, n, int(np.minimum(counts[pol], prm.max_pts).sum()), v[len(v) // 2], v[0],
                     v[-1], n * w * h / v[len(v) // 2] / 1e6), flush=True)
        for pol in policies:
            ctxs, bufs = sets[pol]
            for b in bufs:
                for x in b:
                    x.free()
            for c in ctxs:
                c.close()


if __name__ == "__main__":
    main()
