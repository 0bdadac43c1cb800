#!/usr/bin/env python3
"""Randomised matcher sweep: cusift_match (HIP, through the C ABI) against the CPU oracle over random sizes, both
distances, signed / non-negative / duplicated descriptors.  Run on the GPU box:  python tools/fuzz_match.py [n] [seed]
Bars (tests/test_matching.py): scores within 4e-6, indices equal except near-ties (>= 99 % per case and every
disagreement a near-tie), ambiguity within 1e-4 relative where the indices agree."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cusift_amd import capi  # noqa: E402
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer  # noqa: E402
from oracle_binding import Oracle  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    ctx, oracle = capi.Context(0), Oracle()
    bad = 0
    worst = 0.0
    for case in range(n_cases):
        n1 = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, int(rng.integers(1, 3000))]))
        n2 = int(rng.choice([1, 2, 31, 32, 33, 127, 128, 129, 500, 2047, int(rng.integers(1, 6000))]))
        kind = int(rng.integers(0, 4))

        def pts(n):
            p = np.zeros(n, dtype=SIFT_POINT_DTYPE)
            d = rng.normal(size=(n, 128)).astype(np.float32)
            if kind != 1:
                d = np.abs(d)  # SIFT-like: non-negative
            if kind == 2:
                d = np.round(d * 4) / 4 + 0.125  # coarse values: many exact ties
            p["data"] = d / np.linalg.norm(d, axis=1, keepdims=True)
            p["coords2D"] = rng.uniform(0, 2000, (n, 2)).astype(np.float32)
            return p

        s1, s2 = pts(n1), pts(n2)
        if kind == 3 and n2 > 4:  # duplicates inside image 2 and a copy of image 1's rows
            s2["data"][n2 // 2] = s2["data"][0]
            k = min(n1, n2 // 3)
            s2["data"][-k:] = s1["data"][:k]
        ok = True
        for distance in (1, 0):
            want = s1.copy()
            oracle.match(want, s2, distance)
            d1, d2 = DeviceBuffer.from_numpy(ctx, s1), DeviceBuffer.from_numpy(ctx, s2)
            ctx.match(d1.ptr, n1, d2.ptr, n2, distance)
            ctx.synchronize()
            got = d1.to_numpy(SIFT_POINT_DTYPE, (n1,))
            d1.free()
            d2.free()
            err = float(np.abs(got["score"] - want["score"]).max())
            worst = max(worst, err)
            same = got["match"] == want["match"]
            ok &= err <= 4e-6
            ok &= bool(((got["match"] >= 0) & (got["match"] < n2)).all())
            if not same.all():  # a different index must be a near-tie: its own score within 4e-6 of the oracle's best
                dots = s1["data"][~same].astype(np.float64) @ s2["data"].astype(np.float64).T
                val = 2 - 2 * dots if distance else dots
                mine = val[np.arange(val.shape[0]), got["match"][~same]]
                best = val.min(axis=1) if distance else val.max(axis=1)
                ok &= bool((np.abs(mine - best) <= 8e-6).all())
            if same.any():
                ok &= bool(np.allclose(got["ambiguity"][same], want["ambiguity"][same], rtol=2e-4, atol=6e-5))
        print("%s case %3d: n1=%d n2=%d kind=%d" % ("ok " if ok else "BAD", case, n1, n2, kind))
        bad += not ok
    print("%d of %d cases failed; largest score difference %.3e" % (bad, n_cases, worst))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
