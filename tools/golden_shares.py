#!/usr/bin/env python3
"""The shares the golden gates achieve, for the documentation (CPU; tools/gen_numbers.py reads the result):
the oracle against both of the reference's golden files, plus the orientation tail's diagnosis.

    python tools/golden_shares.py > profiles/r06/golden_shares.json
(The HIP path is bit-identical to the oracle in location, scale and orientation -- tests/test_gpu_parity.py -- and is held
to the same gates on the GPU: test_extract_fixture_vs_reference_golden.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from cusift_amd import synth
    from oracle_binding import Oracle
    from parity_utils import golden_gates, orientation_outliers

    def load(name):
        raw = np.fromfile(os.path.join(ROOT, "tests", "golden", name + ".bin"), dtype=np.uint8)
        n = int(raw[:4].view(np.uint32)[0])
        return raw[4:].view(np.float32).reshape(n, 4)

    prm = dict(num_octaves=6, init_blur=0.0, peak_thresh=0.1, edge_thresh=10.0, lowest_scale=0.0, subsampling=1.0,
               max_pts=16384)
    pts, peaks = Oracle("").extract_with_orientation_peaks(synth.fixture_image(), **prm)
    out = {}
    for name in ("cusift1_check", "cusift1"):
        gold = load(name)
        s = golden_gates(gold, pts, "oracle")
        s["tail"] = orientation_outliers(gold, pts, peaks)
        out[name] = s
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
