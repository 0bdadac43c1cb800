"""Where a wave of describe_all_kernel spends its cycles: run with a library built with -DCUSIFT_EXP=10 (phase stamps
written into unused record fields), on the bench's images.  tools/exp_describe_stamps.sh builds and runs it."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from cusift_amd import synth
from cusift_amd.batch import BatchExtractor

B, w, h = 64, 1920, 1080
ex = BatchExtractor(B, w, h, num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, lowest_scale=0.0,
                    subsampling=1.0, max_pts=4096, tex_frac_bits=8)
imgs = np.stack([synth.tile(1000 + i, w, h, 1.0) for i in range(B)])
d = ex.images_from_numpy(imgs)
for _ in range(3):
    ex.extract(d)
torch.cuda.synchronize()
pts_per_image = ex.to_host()
p = np.concatenate(pts_per_image)
names = [("record + patch staging", p["score"]), ("orientation samples", p["ambiguity"]),
         ("orientation histogram", p["match_xpos"]), ("smoothing + peak", p["match_ypos"]),
         ("descriptor samples", p["match_error"]), ("angle split + zeroing", p["empty"][:, 0]),
         ("gather", p["empty"][:, 1]), ("cell sums + normalisation", p["empty"][:, 2]),
         ("record stores", p["coords3D"][:, 0])]
tot = sum(v.astype(np.float64) for _, v in names)
print("keypoints %d" % len(p))
pre = p["coords3D"][:, 1]
print("first segment: record + geometry %.0f cycles, patch loads + LDS writes %.0f" % (pre.mean(), (p["score"] - pre).mean()))
pw, ph = p["coords3D"][:, 2] // 100, p["coords3D"][:, 2] % 100
for lo, hi in ((0, 24), (24, 32), (32, 41)):
    m = (pw > lo) & (pw <= hi)
    if m.any():
        print("  patch width %2d..%2d: %5.1f%% of keypoints, rows %.1f, staging %.0f cycles" %
              (lo + 1, hi, 100 * m.mean(), ph[m].mean(), (p["score"] - pre)[m].mean()))
print("%-28s %10s %7s %10s %10s" % ("segment", "mean cyc", "share", "median", "p95"))
for n, v in names:
    print("%-28s %10.0f %6.1f%% %10.0f %10.0f" % (n, v.mean(), 100 * v.mean() / tot.mean(), np.median(v),
                                                   np.percentile(v, 95)))
print("%-28s %10.0f" % ("total per keypoint (wave)", tot.mean()))
