#!/usr/bin/env python3
"""What would a lone caller's 64 x 1080p call gain if the library cut it into sub-batches and pipelined them over internal
streams (detection of one sub-batch beside the description of another)?  Emulated with what exists: S streams x (64 / S')
frames per submit, a device-wide wait after every 64 frames -- against one 64-frame call on one stream.  One JSON line."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from cusift_amd import synth
    from cusift_amd.batch import BatchExtractor, PipelinedExtractor

    w, h = 1920, 1080
    kw = dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, lowest_scale=0.0, subsampling=1.0,
              max_pts=32768, tex_frac_bits=8)
    imgs = np.stack([synth.tile(1000 + i, w, h, 1.0) for i in range(64)])
    out = {}
    ex = BatchExtractor(64, w, h, **kw)
    d = ex.images_from_numpy(imgs)
    for _ in range(10):
        ex.extract(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        ex.extract(d)
        torch.cuda.synchronize()   # a lone caller: one call in flight, waited for
    out["one call of 64 frames, one stream, waited for"] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
    ex.close()
    for streams, per in ((2, 32), (4, 16), (2, 16), (4, 8), (3, 16)):
        pipe = PipelinedExtractor(per, w, h, n_streams=streams, n_slots=1, fused_detect=1, **kw)
        parts = [pipe.extractors[0].images_from_numpy(imgs[i:i + per]) for i in range(0, 64, per)]
        for _ in range(5):
            for p in parts:
                pipe.submit(p)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            for p in parts:
                pipe.submit(p)
            torch.cuda.synchronize()
        out["%d sub-batches of %d frames over %d streams, waited for per 64 frames" % (64 // per, per, streams)] = round(
            (time.perf_counter() - t0) / 40 * 1e3, 4)
        for x in pipe.extractors:
            x.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
