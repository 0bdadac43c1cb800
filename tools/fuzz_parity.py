#!/usr/bin/env python3
"""Randomised parity sweep: HIP extraction (through the C ABI) against the CPU oracle over random image sizes,
contents, octave counts, thresholds, initial blurs, capacities and batch sizes.  Run on the GPU box:

    python tools/fuzz_parity.py [n_cases] [seed]

Checks per case, for EVERY keypoint (no tolerated fraction): identical point sets with bit-identical location, scale,
sharpness, edgeness and orientation (NaN in the same places), octave blocks coarsest first, every finite descriptor
within 1e-4 L2 of the oracle's, counts saturate like the reference.  Reports the largest descriptor distance seen."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cusift_amd import capi, synth  # noqa: E402
from oracle_binding import Oracle, pitched  # noqa: E402
from parity_utils import ang_diff, canonical_order  # noqa: E402


def make_image(rng, w, h):
    kind = rng.integers(0, 4)
    if kind == 0:
        img = synth.tile(int(rng.integers(0, 10 ** 6)), w, h, float(rng.choice([0.0, 1.0])))
    elif kind == 1:
        img = synth.blobs(int(rng.integers(0, 10 ** 6)), w, h)
    elif kind == 2:  # smooth noise
        small = rng.uniform(0, 255, size=(max(2, h // 6), max(2, w // 6))).astype(np.float32)
        img = np.kron(small, np.ones((6, 6), dtype=np.float32))[:h, :w]
        img = np.pad(img, ((0, h - img.shape[0]), (0, w - img.shape[1])), mode="edge")
        img = synth.gaussian_blur(img, 1.5) if hasattr(synth, "gaussian_blur") else img
    else:  # white noise, 8 bit
        img = rng.integers(0, 256, size=(h, w)).astype(np.float32)
    return np.ascontiguousarray(img, dtype=np.float32)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    oracle = Oracle()
    bad = 0
    worst = 0.0
    with capi.Context(0) as ctx:
        for case in range(n_cases):
            w = int(rng.choice([4, 8, 12, 36, 64, 100, 124, 240, 244, 256, 320, 484, 500, 640, 964, 1000]))
            if rng.random() < 0.3:
                w += int(rng.integers(1, 4))  # ragged widths: the partial last column group of the fast kernels
            h = int(rng.choice([3, 5, 9, 16, 17, 33, 64, 67, 100, 135, 240, 270, 480, 540]))
            kw = dict(num_octaves=int(rng.integers(1, 7)), init_blur=float(rng.choice([0.0, 0.5, 1.0, 1.3])),
                      peak_thresh=float(rng.choice([0.1, 0.5, 1.0, 3.0])), edge_thresh=float(rng.choice([10.0, 5.0])),
                      lowest_scale=float(rng.choice([0.0, 0.0, 2.0])), subsampling=float(rng.choice([1.0, 1.0, 2.0])),
                      max_pts=int(rng.choice([64, 4096, 32768])))
            fused = int(rng.integers(0, 2))
            n_img = int(rng.choice([1, 1, 2, 5]))
            imgs = [make_image(rng, w, h) for _ in range(n_img)]
            prm = capi.default_params(fused_detect=fused, **kw)
            # the launch sequence is drawn too (results never depend on it): a list per octave or not, the pyramid as a
            # by-product of the detections (every octave / octave 0 only) or the ScaleDown chain first
            lists = int(rng.choice([-1, 1]))
            pyramid = int(rng.choice([-1, 0, 1, 2])) if lists == 1 else -1
            ctx.set_policy(capi.POLICY_OCTAVE_LISTS, lists)
            ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, pyramid)
            stack = np.stack([pitched(i) for i in imgs])
            p = stack.shape[2]
            d_imgs = capi.DeviceBuffer.from_numpy(ctx, stack)
            d_pts = capi.DeviceBuffer(ctx, n_img * prm.max_pts * 588)
            d_pts.zero()
            d_cnt = capi.DeviceBuffer(ctx, 4 * n_img)
            ctx.extract_batch(d_imgs.ptr, n_img, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
            ctx.synchronize()
            counts = d_cnt.to_numpy(np.uint32, (n_img,))
            allpts = d_pts.to_numpy(capi.SIFT_POINT_DTYPE, (n_img, prm.max_pts))
            msg = []
            for i, img in enumerate(imgs):
                big = dict(kw)
                big["max_pts"] = 1 << 19
                want_all = oracle.extract(img, **big)
                if int(counts[i]) != len(want_all):
                    msg.append("img %d: count %d, oracle %d" % (i, counts[i], len(want_all)))
                    continue
                n = min(int(counts[i]), prm.max_pts)
                got = allpts[i, :n]
                if len(want_all) > prm.max_pts:
                    # saturated: the coarse-octave blocks that fit entirely are complete; just check membership
                    wkeys = {(round(float(a), 3), round(float(b), 3), round(float(c), 3))
                             for a, b, c in zip(want_all["coords2D"][:, 0], want_all["coords2D"][:, 1], want_all["scale"])}
                    miss = sum((round(float(a), 3), round(float(b), 3), round(float(c), 3)) not in wkeys
                               for a, b, c in zip(got["coords2D"][:, 0], got["coords2D"][:, 1], got["scale"]))
                    if miss > 0.01 * n + 1:
                        msg.append("img %d: saturated run holds %d points the oracle does not have" % (i, miss))
                    continue
                if n == 0:
                    continue
                if np.any(np.diff(got["subsampling"]) > 0):
                    msg.append("img %d: octave blocks not coarsest first" % i)
                a, b = canonical_order(want_all), canonical_order(got)
                for f in ("subsampling", "coords2D", "scale", "sharpness", "edgeness", "orientation"):
                    if not np.array_equal(a[f], b[f], equal_nan=True):
                        msg.append("img %d: %s differs (%d of %d)" % (i, f, int((a[f] != b[f]).sum()), a[f].size))
                fin = np.isfinite(a["data"]).all(axis=1)
                if not np.array_equal(np.isfinite(b["data"]).all(axis=1), fin):
                    msg.append("img %d: NaN-descriptor pattern differs" % i)
                elif fin.any():
                    l2 = np.linalg.norm(a["data"][fin].astype(np.float64) - b["data"][fin].astype(np.float64), axis=1)
                    worst = max(worst, float(l2.max()))
                    if l2.max() >= 1e-4:
                        msg.append("img %d: %d descriptors >= 1e-4 (max %.2e)" % (i, int((l2 >= 1e-4).sum()), l2.max()))
            status = "ok " if not msg else "BAD"
            bad += bool(msg)
            print("%s case %3d: %dx%d x%d oct=%d blur=%.1f thr=%.1f edge=%.0f low=%.0f sub=%.0f max=%d fused=%d lists=%d pyr=%d counts=%s %s"
                  % (status, case, w, h, n_img, kw["num_octaves"], kw["init_blur"], kw["peak_thresh"], kw["edge_thresh"],
                     kw["lowest_scale"], kw["subsampling"], kw["max_pts"], fused, lists, pyramid, list(map(int, counts)),
                     "; ".join(msg)),
                  flush=True)
            for bfr in (d_imgs, d_pts, d_cnt):
                bfr.free()
    print("%d of %d cases failed; largest descriptor L2 distance to the oracle over all keypoints: %.3e" % (bad, n_cases, worst))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
