#!/usr/bin/env python3
"""BASELINE configs[1]: a single 1920x1080 image, 5 octaves, initBlur=1.0, thresh=3.0 on one MI355X.

Latency of one extraction (device-resident image -> SiftData in HBM, host notified), eager launches versus the
recorded hipGraph (cusift_graph_*), the back-to-back rate without a host wait per frame, and host image in -> host
SiftData out.  No torch: the C ABI only.  The measurement is bench.py's `config_legs["configs[1]"]`
(bench_legs/configs.py: single_frame); this tool prints it alone, as one JSON line.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(w=1920, h=1080, iters=200, warmup=20):
    from bench_legs.configs import single_frame
    from cusift_amd import capi, synth

    return single_frame(capi, synth, 0, w=w, h=h, iters=iters, warmup=warmup)


if __name__ == "__main__":
    print(json.dumps(measure(iters=int(os.environ.get("LAT_ITERS", "200")))))
