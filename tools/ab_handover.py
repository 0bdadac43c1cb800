#!/usr/bin/env python3
"""Round 6, VERDICT item 3: the upper bound of an in-launch octave hand-over for a LONE caller (one batch in flight).

    CUSIFT_AMD_LIB=cusift_amd/libcusift_amd_lab.so [CUSIFT_UNORDERED_COARSE=1] python tools/ab_handover.py

One process = one arm (the lab library reads the knob when a context is created).  With CUSIFT_UNORDERED_COARSE the launches
behind octave 0's detection run on a second stream WITHOUT waiting for it (they read the previous call's octave-1 image:
the same pixels in this loop) -- perfect overlap at no synchronisation cost, i.e. what no real hand-over can beat.
Prints one JSON line: ms per call for 64 x 1080p and for one 8192 x 8192 image."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from cusift_amd import capi, synth
    from cusift_amd.batch import BatchExtractor

    out = {"arm": "unordered coarse octaves (upper bound)" if os.environ.get("CUSIFT_UNORDERED_COARSE") else "product order",
           "library": os.path.basename(capi.lib()._name)}
    for name, n, w, h, max_pts, reps in (("64 x 1080p", 64, 1920, 1080, 32768, 60), ("1 x 8192^2", 1, 8192, 8192, 1 << 19, 60)):
        prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=max_pts)
        ex = BatchExtractor(n, w, h, params=prm)
        # (a lone caller's default for these sizes is "octave 0 only"; the 8192^2 image is below the 64-million-pixel limit
        # only just above -- force the policy so that both sizes run the same sequence)
        ex.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, 1)
        if n == 1:
            imgs = synth.tile(4242, w, h, preblur=1.0)[None]
        else:
            imgs = np.stack([synth.tile(1000 + i, w, h, 1.0) for i in range(n)])
        d = ex.images_from_numpy(imgs)
        for _ in range(20):
            ex.extract(d)
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(reps):
                ex.extract(d)
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / reps * 1e3)
        out[name] = {"ms_per_call_median_of_5": round(sorted(best)[2], 4), "min": round(min(best), 4),
                     "keypoints": int(ex.valid_counts().sum().item())}
        ex.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
