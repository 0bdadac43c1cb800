"""BASELINE.json configs[2] and configs[4] at their REAL sizes on one MI355X, checked -- not just timed.

configs[2]  batch of 64 x 1920x1080 (5 octaves, initBlur 1.0, thresh 3.0): exactly the batch bench.py times, through
            cusift_extract_batch.  The arena of this size (cuSIFT.cu:74-98 sizes the reference's per image) crosses
            2^31-byte offsets that an 8-image batch never touches.  Sampled images are compared with the CPU oracle
            keypoint by keypoint; all 64 counts must equal the single-image extractions.
configs[4]  one 8192x8192 image over 8 ranks (strip tiling + halo exchange): the whole-image extraction against the CPU
            oracle keypoint by keypoint, then 8 virtual ranks on one GPU whose SiftData united must equal it bit for bit.
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


def test_whole_8192_matches_oracle(ctx, oracle):
    """configs[4]'s image at full size against the ORACLE, every keypoint (the strip test below compares the tiled result
    with this whole-image extraction, i.e. the HIP path with itself: until round 4 the largest oracle-checked image was
    4096 x 3072).  The oracle needs ~15 s for the 67 Mpx on one host core."""
    W = H = 8192
    img = synth.tile(4242, W, H, preblur=1.0)
    kw = dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=1 << 18)
    want = oracle.extract(img, **kw)
    prm = capi.default_params(**kw)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
    d_pts.free()
    assert 50000 < n == len(want) < prm.max_pts
    assert np.all(np.diff(h_pts[:n]["subsampling"]) <= 0)
    compare_sets(want, h_pts[:n])


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


def _extract_batch(ctx, stack, prm, canary_records=64):
    """cusift_extract_batch over `stack` (n, h, w); returns (raw counters, records [n, max_pts], canary intact?)."""
    n, h, w = stack.shape
    p = capi.ialign_up(w, 128)
    d_imgs = DeviceBuffer.from_numpy(ctx, pitched(stack, p) if p != w else stack)
    # the output block is followed by `canary_records` records of 0xA5: nothing may be written past slot max_pts - 1
    # of the last image (and, by the same indexing, of any image: image i + 1's records would be clobbered)
    d_pts = DeviceBuffer(ctx, (n * prm.max_pts + canary_records) * 588)
    ctx.memset(d_pts.ptr, 0xA5, d_pts.nbytes)
    d_cnt = DeviceBuffer(ctx, 4 * n)
    ctx.extract_batch(d_imgs.ptr, n, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    cnt = d_cnt.to_numpy(np.uint32, (n,))
    raw = d_pts.to_numpy(np.uint8, (n * prm.max_pts + canary_records, 588))
    canary_ok = bool((raw[n * prm.max_pts:] == 0xA5).all())
    rec = raw[: n * prm.max_pts].copy().view(SIFT_POINT_DTYPE).reshape(n, prm.max_pts)
    for b in (d_imgs, d_pts, d_cnt):
        b.free()
    return cnt, rec, canary_ok


def test_blobs_batch8_1080p_matches_oracle(ctx, oracle):
    """The `blobs` content bench.py times (SURVEY.md section 8d's secondary generator, 9-10 k keypoints per image, 3.5 x
    the headline content): a batch of 8 x 1080p through cusift_extract_batch against the oracle, every keypoint."""
    n, w, h = 8, 1920, 1080
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        host = list(pool.map(lambda s: synth.blobs(s, w, h), [1000 + i for i in range(n)]))
        prm = capi.default_params(**BENCH_KW)
        kw = dict(BENCH_KW)
        want = list(pool.map(lambda im: oracle.extract(im, **kw), host))
    cnt, rec, canary_ok = _extract_batch(ctx, np.stack(host), prm)
    assert canary_ok
    assert cnt.min() > 3000 and cnt.max() < prm.max_pts
    for i in range(n):
        got = rec[i, : cnt[i]]
        assert np.all(np.diff(got["subsampling"]) <= 0)
        compare_sets(want[i], got)


def test_bench_images_with_initblur0_match_oracle(ctx, oracle):
    """bench.py's `initblur0` leg: the headline images with initBlur = 0.0 declared (the only value the reference's own
    test uses, test/detector.cpp:43) -- no identity levels in octave 0.  Two images of the batch, every keypoint."""
    w, h = 1920, 1080
    kw = dict(BENCH_KW, init_blur=0.0)
    prm = capi.default_params(**kw)
    host = [synth.tile(1000 + i, w, h, 1.0) for i in (0, 37)]
    cnt, rec, canary_ok = _extract_batch(ctx, np.stack(host), prm)
    assert canary_ok and cnt.max() < prm.max_pts
    for i, im in enumerate(host):
        compare_sets(oracle.extract(im, **kw), rec[i, : cnt[i]])


def test_counter_overflow_batch_is_a_subset_of_the_oracle(ctx, oracle):
    """bench.py's raw-tile leg: un-pre-blurred tiles with initBlur = 1.0 declared saturate maxPts = 32768 on every image.
    The reference truncates (numPts = min(counter, maxPts), cuSIFT.cu:110) after letting every overflowing append land
    in slot maxPts - 1 (cuSIFT_D.cu:512-514); this build drops them.  Demanded: the counter keeps counting, exactly
    maxPts records are valid, the coarse-octave blocks are complete and equal the oracle's keypoint by keypoint, the
    octave-0 rows that fitted are a SUBSET of the oracle's octave-0 keypoints (each with the oracle's values), and
    nothing is written past slot maxPts - 1."""
    n, w, h = 2, 1920, 1080
    host = [synth.tile(1000 + i, w, h, 0.0) for i in range(n)]
    prm = capi.default_params(**BENCH_KW)
    cnt, rec, canary_ok = _extract_batch(ctx, np.stack(host), prm)
    assert canary_ok, "records were written past the end of the output block"
    big = dict(BENCH_KW, max_pts=400000)  # the oracle with room for everything
    for i in range(n):
        want = oracle.extract(host[i], **big)
        assert len(want) > prm.max_pts, "the content no longer overflows: pick another"
        assert int(cnt[i]) == len(want), (int(cnt[i]), len(want))  # the counter counts every accepted keypoint
        got = rec[i]  # all max_pts slots are valid records
        assert np.all(np.diff(got["subsampling"]) <= 0)  # coarsest octave first, octave 0 last
        coarse_w, coarse_g = want[want["subsampling"] > 1.0], got[got["subsampling"] > 1.0]
        compare_sets(coarse_w, coarse_g)  # octaves 4..1: complete
        fine_g = got[got["subsampling"] == 1.0]
        assert len(coarse_g) + len(fine_g) == prm.max_pts and len(fine_g) > 1000
        # octave-0 rows: distinct, and each is one of the oracle's octave-0 keypoints with the oracle's values
        fine_w = want[want["subsampling"] == 1.0]
        key = lambda p: np.ascontiguousarray(np.stack([p["coords2D"][:, 0], p["coords2D"][:, 1], p["scale"]], 1)).view(  # noqa: E731
            np.dtype((np.void, 12))).ravel()
        kw_, kg_ = key(fine_w), key(fine_g)
        assert len(np.unique(kg_)) == len(kg_)
        order = np.argsort(kw_)
        pos = np.searchsorted(kw_[order], kg_)
        assert (pos < len(kw_)).all() and (kw_[order][np.minimum(pos, len(kw_) - 1)] == kg_).all()
        compare_sets(fine_w[order][pos], fine_g)


def test_strips_8192_distributed_over_the_transport(ctx):
    """configs[4] at full size with the halos and the collapse gather moved by cusift_exchange_halos / cusift_exchange_rows
    between 8 RANKS (threads, one context + communicator each, tests/fake_rccl) instead of virtual ranks' device copies:
    cusift_tiled_extract on every rank, then the all-gatherv merge; every rank's merged SiftData == the whole image."""
    from cusift_amd.dist import SiftGatherer
    from cusift_amd.tiling import run_distributed
    from test_multirank_gpu import _same_extracted, run_ranks

    W = H = 8192
    P, n_oct = 8, 7
    img = synth.tile(4242, W, H, preblur=1.0)
    prm = capi.default_params(num_octaves=n_oct, init_blur=1.0, peak_thresh=3.0, max_pts=1 << 18)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
    want = canonical_order(h_pts[:n])
    d_pts.free()
    dev = torch.device("cuda", 0)
    full = torch.from_numpy(img).to(dev)
    torch.cuda.synchronize()
    rows = H // P
    sprm = capi.default_params(num_octaves=n_oct, init_blur=1.0, peak_thresh=3.0, max_pts=1 << 16)

    def rank_fn(rank, make_comm):
        with torch.cuda.device(dev):
            c = capi.Context(0)
            comm = make_comm(c)
            ext = StripExtractor(rank, P, W, H, sprm, device=dev, comm=comm)
            pts, cnt = run_distributed(ext, full[rank * rows:(rank + 1) * rows])
            ext.check()
            g = SiftGatherer(comm, 1, sprm.max_pts, region_cap=sprm.max_pts, device=dev)
            counts, gathered, totals = g.gather(pts, cnt)
            c.synchronize()
            merged = None
            if rank in (0, 5):  # two ranks bring their merged copy to the host (8 x 150 MB otherwise)
                merged = np.concatenate([x.cpu().numpy() for x in SiftGatherer.regions(gathered, totals)])
                merged = merged.view(SIFT_POINT_DTYPE).reshape(-1)
            ext.close()
            comm.close()
            c.close()
            return merged, [int(t) for t in totals]

    res = run_ranks(P, rank_fn, timeout=600)
    assert all(r[1] == res[0][1] for r in res) and sum(res[0][1]) == n
    for k in (0, 5):
        assert _same_extracted(canonical_order(res[k][0]), want)
