import os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests"); sys.path.insert(0, ROOT+"/tools")
from cusift_amd import capi
from oracle_binding import Oracle, pitched
from parity_utils import ang_diff, canonical_order
import fuzz_parity as F
rng = np.random.default_rng(1)
oracle = Oracle()
# replay the generator up to case 88
target = int(sys.argv[1]) if len(sys.argv) > 1 else 88
with capi.Context(0) as ctx:
    for case in range(target + 1):
        w = int(rng.choice([4, 8, 12, 36, 64, 100, 124, 240, 244, 256, 320, 484, 500, 640, 964, 1000]))
        if rng.random() < 0.3:
            w += int(rng.integers(1, 4))
        h = int(rng.choice([3, 5, 9, 16, 17, 33, 64, 67, 100, 135, 240, 270, 480, 540]))
        kw = dict(num_octaves=int(rng.integers(1, 7)), init_blur=float(rng.choice([0.0, 0.5, 1.0, 1.3])),
                  peak_thresh=float(rng.choice([0.1, 0.5, 1.0, 3.0])), edge_thresh=float(rng.choice([10.0, 5.0])),
                  lowest_scale=float(rng.choice([0.0, 0.0, 2.0])), subsampling=float(rng.choice([1.0, 1.0, 2.0])),
                  max_pts=int(rng.choice([64, 4096, 32768])))
        fused = int(rng.integers(0, 2))
        n_img = int(rng.choice([1, 1, 2, 5]))
        imgs = [F.make_image(rng, w, h) for _ in range(n_img)]
    img = imgs[0]
    print("case", target, w, h, kw, "img std", img.std(), "unique", len(np.unique(img)))
    for bits in (8, 0):
        k2 = dict(kw); k2["max_pts"] = 32768
        want = oracle.extract(img, tex_frac_bits=bits, **k2)
        prm = capi.default_params(fused_detect=fused, tex_frac_bits=bits, **k2)
        d_pts = capi.DeviceBuffer(ctx, prm.max_pts * 588)
        h_pts = np.zeros(prm.max_pts, dtype=capi.SIFT_POINT_DTYPE)
        n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
        a, b = canonical_order(want), canonical_order(h_pts[:n])
        dor = ang_diff(a["orientation"].astype(np.float64), b["orientation"].astype(np.float64))
        ok = dor < 1e-3
        l2 = np.linalg.norm(a["data"][ok].astype(np.float64) - b["data"][ok].astype(np.float64), axis=1)
        print("bits", bits, "n", n, "ori ok", ok.mean(), "desc<1e-4", (l2 < 1e-4).mean(), "max", l2.max(), "p99.9", np.percentile(l2, 99.9),
              "scale range", a["scale"].min(), a["scale"].max())
        badl = l2[l2 >= 1e-4]
        print("   bad l2 quantiles", np.percentile(badl, [10, 50, 90]) if len(badl) else None)
