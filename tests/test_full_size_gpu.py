"""BASELINE.json configs[2] and configs[4] at their REAL sizes on one MI355X, checked -- not just timed.

configs[2]  batch of 64 x 1920x1080 (5 octaves, initBlur 1.0, thresh 3.0): exactly the batch bench.py times, through
            cusift_extract_batch.  The arena of this size (cuSIFT.cu:74-98 sizes the reference's per image) crosses
            2^31-byte offsets that an 8-image batch never touches.  Sampled images are compared with the CPU oracle
            keypoint by keypoint; all 64 counts must equal the single-image extractions.
configs[4]  one 8192x8192 image over 8 ranks (strip tiling + halo exchange): 8 virtual ranks on one GPU, the union of
            their SiftData must equal the whole-image extraction bit for bit.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

from cusift_amd import capi, synth
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer
from cusift_amd.tiling import StripExtractor, run_virtual
from oracle_binding import pitched
from parity_utils import canonical_order
from test_gpu_parity import compare_sets

pytestmark = pytest.mark.gpu

BENCH_KW = dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, lowest_scale=0.0, subsampling=1.0,
                max_pts=32768, tex_frac_bits=8)


def test_batch64_1080p_matches_oracle(ctx, oracle):
    n, w, h = 64, 1920, 1080
    seeds = [1000 + i for i in range(n)]  # bench.py's images of rank 0
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        host = list(pool.map(lambda s: synth.tile(s, w, h, 1.0), seeds))
    p = capi.ialign_up(w, 128)
    assert p == w
    stack = np.stack(host)
    prm = capi.default_params(**BENCH_KW)
    d_imgs = DeviceBuffer.from_numpy(ctx, stack)
    d_pts = DeviceBuffer(ctx, n * prm.max_pts * 588)
    d_cnt = DeviceBuffer(ctx, 4 * n)
    assert d_pts.nbytes > (1 << 30)
    ctx.extract_batch(d_imgs.ptr, n, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    cnt = d_cnt.to_numpy(np.uint32, (n,))
    assert cnt.min() > 1000 and cnt.max() < prm.max_pts
    assert ctx.arena_bytes() > 150 << 20  # octaves 1..4 of 64 images (the fused detection holds no DoG block)

    # every image of the batch == that image extracted alone (counts), through the blocking single-image driver
    with capi.Context(0) as single:
        d_one = DeviceBuffer(single, prm.max_pts * 588)
        for i in range(n):
            k = single.extract(d_imgs.ptr + i * h * p * 4, w, h, p, prm, d_one.ptr, None)
            assert k == int(cnt[i]), (i, k, int(cnt[i]))
        d_one.free()

    # sampled images against the oracle, every keypoint
    rec = np.empty(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    for i in (0, 21, 42, 63):
        ctx.d2h(rec, d_pts.ptr + i * prm.max_pts * 588)
        got = rec[: cnt[i]].copy()
        assert np.all(np.diff(got["subsampling"]) <= 0)  # octave blocks coarsest first (cuSIFT.cu:190-196)
        want = oracle.extract(host[i], **BENCH_KW)
        compare_sets(want, got)
    for b in (d_imgs, d_pts, d_cnt):
        b.free()


@pytest.mark.parametrize("n_oct", [5, 7])
def test_strips_8192_equal_whole_image(ctx, n_oct):
    """5 octaves: all tiled.  7 octaves: octaves 5 and 6 (32 and 16 owned rows per rank < the 48-row halo) collapse onto
    rank 0, which runs the whole-image driver on the gathered 256-row octave."""
    W = H = 8192
    P = 8
    img = synth.tile(4242, W, H, preblur=1.0)
    prm = capi.default_params(num_octaves=n_oct, init_blur=1.0, peak_thresh=3.0, max_pts=1 << 18)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
    want = canonical_order(h_pts[:n])
    d_pts.free()
    assert 50000 < n < prm.max_pts
    dev = torch.device("cuda", 0)
    full = torch.from_numpy(img).to(dev)
    rows = H // P
    sprm = capi.default_params(num_octaves=n_oct, init_blur=1.0, peak_thresh=3.0, max_pts=1 << 16)
    exts = [StripExtractor(k, P, W, H, sprm, device=dev) for k in range(P)]
    assert exts[0].plan.collapse == 5
    parts = run_virtual(exts, [full[k * rows:(k + 1) * rows] for k in range(P)])
    for k, pts in enumerate(parts):
        assert 0 < len(pts) < sprm.max_pts
        assert np.all(np.diff(pts["subsampling"]) <= 0)
    got = canonical_order(np.concatenate(parts))
    assert len(got) == len(want)
    for f in ("subsampling", "coords2D", "scale", "sharpness", "edgeness", "orientation", "data"):
        np.testing.assert_array_equal(got[f], want[f], err_msg=f)
    for e in exts:
        e.close()
