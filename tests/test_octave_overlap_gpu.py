"""The staged drivers of cusift_extract_batch: octave 0 beside the coarser octaves (a lone caller:
cusift_params.concurrent_batches < 2), every octave's keypoints to a list of its own (whenever the lists fit), or both.

A staged driver -- detections appending record heads to staging lists in the arena, joined and moved into place by
describe_all_kernel -- must leave the same SiftData as the plain one-stream driver: the same records (bit for bit, as sets per
octave: the order inside an octave is unspecified in both and in the reference), coarsest octave first, the same counter
(it keeps counting beyond max_pts), and under saturation the coarser octaves survive whole, as in the reference
(cuSIFT.cu:190-196 searches them first).  The default `ctx` of the other parity tests forks too, so every whole-image
test of test_gpu_parity.py checks the forked driver against the oracle; here the two drivers face each other.
"""
import numpy as np
import pytest

from cusift_amd import capi, synth
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer
from oracle_binding import pitched
from parity_utils import canonical_order

pytestmark = pytest.mark.gpu

FIELDS = ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data")


def images(n, w, h, blur=1.0):
    return [synth.tile(4000 + i, w, h, blur) for i in range(n)]


def run_batch(ctx, imgs, prm, fill=0):
    n = len(imgs)
    h, w = imgs[0].shape
    stack = np.stack([pitched(i) for i in imgs])
    p = stack.shape[2]
    d_imgs = DeviceBuffer.from_numpy(ctx, stack)
    d_pts = DeviceBuffer(ctx, n * prm.max_pts * 588)
    if fill:
        ctx.memset(d_pts.ptr, fill, n * prm.max_pts * 588)
    else:
        d_pts.zero()
    d_cnt = DeviceBuffer(ctx, 4 * n)
    ctx.extract_batch(d_imgs.ptr, n, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    cnt = d_cnt.to_numpy(np.uint32, (n,)).copy()
    pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (n, prm.max_pts)).copy()
    for b in (d_imgs, d_pts, d_cnt):
        b.free()
    return cnt, pts


def same_records(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f], equal_nan=True) for f in FIELDS)


def context_with(overlap, stage_all=None, no_multi=False, pyramid=0):
    """A context with the launch policy forced (cusift_ctx_set_policy)."""
    c = capi.Context(0)
    c.set_policy(capi.POLICY_SIDE_STREAM, overlap)
    if stage_all is not None:
        c.set_policy(capi.POLICY_OCTAVE_LISTS, stage_all)
    if no_multi:
        c.set_policy(capi.POLICY_LAUNCH_PER_OCTAVE, 1)
    c.set_policy(capi.POLICY_PYRAMID_IN_DETECT, pyramid)
    return c


# how the keypoints of a call reach SiftData (cusift_extract_batch): octave 0 on the side stream with a list of its own
# and the coarser octaves in place; every octave to a list of its own and the coarser octaves searched by one launch
# (or by a launch each); both
# ...; the pyramid as a by-product of the detections, searched finest first (every octave / octave 0 only, then the
# ScaleDown chain and one launch for the coarser octaves / then a launch per octave)
MODES = {"fork": (3, 0), "lists": (0, 1), "lists, a launch per octave": (0, 1, True), "fork+lists": (3, 1),
         "pyramid in the detections": (0, 1, False, 2), "pyramid in octave 0's detection": (0, 1, False, 1),
         "pyramid in octave 0's detection, a launch per octave": (0, 1, True, 1)}


@pytest.fixture(params=sorted(MODES))
def ctx(request):
    """This module's `ctx`: one of the staged drivers, forced whatever the size of the call."""
    c = context_with(*MODES[request.param])
    c.mode = request.param
    yield c
    c.close()


@pytest.fixture
def one_stream():
    """The plain driver: one stream, every keypoint appended in place."""
    c = context_with(0, 0)
    yield c
    c.close()


@pytest.mark.parametrize("n,w,h,n_oct,blur", [(3, 640, 480, 5, 1.0), (1, 1920, 1080, 5, 1.0), (2, 333, 257, 6, 0.0),
                                              (5, 128, 96, 3, 0.5), (1, 64, 48, 2, 0.0)])
def test_forked_equals_one_stream(ctx, one_stream, n, w, h, n_oct, blur):
    prm = capi.default_params(num_octaves=n_oct, init_blur=blur, peak_thresh=2.0, max_pts=16384)
    imgs = images(n, w, h, blur)
    cnt_f, pts_f = run_batch(ctx, imgs, prm, fill=0x5A)
    cnt_s, pts_s = run_batch(one_stream, imgs, prm, fill=0x5A)
    assert ctx.forks() == (1 if "fork" in ctx.mode else 0) and one_stream.forks() == 0
    np.testing.assert_array_equal(cnt_f, cnt_s)
    assert w < 128 or cnt_f.sum() > 20 * n  # (a 64 x 48 tile may hold no keypoint at all)
    for i in range(n):
        a, b = pts_f[i, : cnt_f[i]], pts_s[i, : cnt_s[i]]
        for lst in (a, b):  # coarsest octave first in both
            assert np.all(np.diff(lst["subsampling"]) <= 0)
        assert same_records(canonical_order(a), canonical_order(b))
        # nothing but the fields extraction writes was touched: the staged heads carry no other field into the record
        for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos", "match_error", "empty", "coords3D"):
            assert np.all(np.ascontiguousarray(a[f]).view(np.uint8) == 0x5A), f
            assert np.all(np.ascontiguousarray(b[f]).view(np.uint8) == 0x5A), f
        # ... nor anything beyond the list
        tail_f = pts_f[i, cnt_f[i]:].view(np.uint8)
        assert np.all(tail_f == 0x5A)


def test_saturation_keeps_the_coarser_octaves(ctx, one_stream):
    """max_pts smaller than the image's keypoints: the counter counts on, the list holds max_pts records, every keypoint of
    the coarser octaves among them, the rest of the room filled with octave 0's -- in both drivers."""
    imgs = images(2, 640, 480)
    big = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=2.0, max_pts=16384)
    cnt_all, pts_all = run_batch(one_stream, imgs, big)
    coarse = [int((pts_all[i, : cnt_all[i]]["subsampling"] > 1.0).sum()) for i in range(2)]
    assert min(coarse) > 30 and min(cnt_all) > 4 * max(coarse) // 3
    cap = max(coarse) + 40
    small = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=2.0, max_pts=cap)
    for c in (ctx, one_stream):
        cnt, pts = run_batch(c, imgs, small)
        np.testing.assert_array_equal(cnt, cnt_all)  # keeps counting (cuSIFT_D.cu:512 atomicInc with no bound of its own)
        for i in range(2):
            got = pts[i]
            assert np.all(np.diff(got["subsampling"]) <= 0)
            whole = canonical_order(pts_all[i, : cnt_all[i]])
            want_coarse = whole[whole["subsampling"] > 1.0]
            got_coarse = canonical_order(got[got["subsampling"] > 1.0])
            assert same_records(got_coarse, want_coarse)
            fine = got[got["subsampling"] == 1.0]
            assert len(fine) == cap - coarse[i]
            # every octave-0 record is one of the image's octave-0 keypoints, none twice
            pool = whole[whole["subsampling"] == 1.0]
            key = lambda r: r[["coords2D", "scale", "orientation"]].tobytes()  # noqa: E731
            have = {key(r) for r in pool}
            seen = {key(r) for r in fine}
            assert len(seen) == len(fine) and seen <= have


def test_coarse_list_alone_fills_the_record_array(ctx, one_stream):
    """max_pts below even the coarser octaves' count: octave 0 gets no room at all."""
    imgs = images(1, 640, 480)
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=2.0, max_pts=25)
    cnt_f, pts_f = run_batch(ctx, imgs, prm)
    cnt_s, pts_s = run_batch(one_stream, imgs, prm)
    np.testing.assert_array_equal(cnt_f, cnt_s)
    assert np.all(pts_f[0]["subsampling"] > 1.0) and np.all(pts_s[0]["subsampling"] > 1.0)


@pytest.mark.parametrize("kw", [dict(lowest_scale=2.0), dict(lowest_scale=40.0), dict(num_octaves=1),
                                dict(subsampling=2.0), dict(root_sift=1), dict(concurrent_batches=4)])
def test_cases_without_a_fork_and_with_other_parameters(ctx, one_stream, kw):
    """lowest_scale drops octave 0 (or all but the coarsest) from the search, one octave leaves nothing to fork from, a
    pipelining caller (concurrent_batches >= 2) keeps the one-stream driver: all equal the one-stream context."""
    base = dict(num_octaves=5, init_blur=1.0, peak_thresh=2.0, max_pts=8192)
    base.update(kw)
    prm = capi.default_params(**base)
    imgs = images(2, 480, 360)
    cnt_f, pts_f = run_batch(ctx, imgs, prm)
    cnt_s, pts_s = run_batch(one_stream, imgs, prm)
    np.testing.assert_array_equal(cnt_f, cnt_s)
    for i in range(2):
        assert same_records(canonical_order(pts_f[i, : cnt_f[i]]), canonical_order(pts_s[i, : cnt_s[i]]))


def test_host_entry_point_and_graph_replay(ctx, one_stream):
    """cusift_extract_host keeps its upload image clear of the staging list; a recorded graph holds the fork and the join."""
    img = images(1, 800, 600)[0]
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=2.0, max_pts=8192)
    outs = []
    for c in (ctx, one_stream):
        d_pts = DeviceBuffer(c, prm.max_pts * 588)
        h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
        n = c.extract_host(img, prm, d_pts.ptr, h_pts)
        outs.append(canonical_order(h_pts[:n]))
        d_pts.free()
    assert len(outs[0]) > 300 and same_records(outs[0], outs[1])

    src = pitched(img)
    h, w = img.shape
    p = src.shape[1]
    with context_with(*MODES[ctx.mode]) as c:
        d_img = DeviceBuffer.from_numpy(c, src)
        d_pts = DeviceBuffer(c, prm.max_pts * 588)
        d_cnt = DeviceBuffer(c, 4)
        g = c.record_graph(d_img.ptr, 1, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
        assert c.forks() == (1 if "fork" in ctx.mode else 0)  # the recording holds the fork and the join
        for _ in range(3):
            d_pts.zero()
            g.launch()
            c.synchronize()
            n = int(d_cnt.to_numpy(np.uint32, (1,))[0])
            got = d_pts.to_numpy(SIFT_POINT_DTYPE, (prm.max_pts,))[:n]
            assert np.all(np.diff(got["subsampling"]) <= 0)
            assert same_records(canonical_order(got), outs[1])
        g.close()
        for b in (d_img, d_pts, d_cnt):
            b.free()


def test_back_to_back_calls_do_not_race_on_the_staging_list(ctx, one_stream):
    """Consecutive extractions on one context reuse the staging list: the next call's fork waits for the previous call's
    describe (stream order through the fork event)."""
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=2.0, max_pts=8192)
    sets = [images(2, 640, 480), [i[::-1].copy() for i in images(2, 640, 480)], [np.roll(i, 91, 1) for i in images(2, 640, 480)]]
    h, w = sets[0][0].shape
    bufs = []
    for imgs in sets:  # enqueue all three without a synchronisation in between
        stack = np.stack([pitched(i) for i in imgs])
        d_imgs = DeviceBuffer.from_numpy(ctx, stack)
        d_pts = DeviceBuffer(ctx, 2 * prm.max_pts * 588)
        d_cnt = DeviceBuffer(ctx, 8)
        bufs.append((d_imgs, d_pts, d_cnt, stack.shape[2]))
    for _ in range(3):
        for d_imgs, d_pts, d_cnt, p in bufs:
            ctx.extract_batch(d_imgs.ptr, 2, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    for imgs, (d_imgs, d_pts, d_cnt, p) in zip(sets, bufs):
        cnt = d_cnt.to_numpy(np.uint32, (2,))
        pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (2, prm.max_pts))
        cnt_s, pts_s = run_batch(one_stream, imgs, prm)
        np.testing.assert_array_equal(cnt, cnt_s)
        for i in range(2):
            assert same_records(canonical_order(pts[i, : cnt[i]]), canonical_order(pts_s[i, : cnt_s[i]]))
        for b in (d_imgs, d_pts, d_cnt):
            b.free()


def test_list_counters_left_clean_survive_any_interleaving(ctx, one_stream):
    """join_counts_kernel leaves the lists' counters zero and the next extraction of the context skips its memset
    (cusift_extract_batch: seg_clean).  That bookkeeping must hold whatever is called in between: other geometries (the
    counters move inside the arena), a recorded graph's replays, a policy change, an arena that grows."""
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=2.0, max_pts=4096)
    small = images(2, 320, 240)
    other = [np.roll(i, 37, 0) for i in images(3, 320, 240)]
    large = images(1, 800, 600)
    want = {k: run_batch(one_stream, v, prm) for k, v in (("small", small), ("other", other), ("large", large))}

    def check(name, imgs):
        cnt, pts = run_batch(ctx, imgs, prm)
        np.testing.assert_array_equal(cnt, want[name][0], err_msg=name)
        for i in range(len(imgs)):
            assert same_records(canonical_order(pts[i, : cnt[i]]), canonical_order(want[name][1][i, : cnt[i]])), name

    check("small", small)
    check("small", small)  # the second call finds clean counters
    check("other", other)  # three images: the counters sit elsewhere
    check("small", small)
    src = np.stack([pitched(i) for i in small])
    d_imgs = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer(ctx, 2 * prm.max_pts * 588)
    d_cnt = DeviceBuffer(ctx, 8)
    g = ctx.record_graph(d_imgs.ptr, 2, 320, 240, src.shape[2], 240 * src.shape[2], prm, d_pts.ptr, d_cnt.ptr)
    check("small", small)  # a recording runs nothing: it must not be taken for a join that left the counters clean
    check("other", other)
    for _ in range(2):
        g.launch()
        ctx.synchronize()
        np.testing.assert_array_equal(d_cnt.to_numpy(np.uint32, (2,)), want["small"][0])
        check("small", small)  # an eager call between the replays
    g.close()
    for b in (d_imgs, d_pts, d_cnt):
        b.free()
    check("large", large)  # the arena grows: everything moves
    check("small", small)
    old = ctx.get_policy(capi.POLICY_PYRAMID_IN_DETECT)
    ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, 0 if old else 2)
    if ctx.get_policy(capi.POLICY_OCTAVE_LISTS) == 1:
        check("small", small)
        check("small", small)
    ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, old)
    check("small", small)


def test_default_policy(one_stream):
    """A context never forks unless asked to (round 4: the side stream is opt-in, nothing is decided by timing).  Asked
    (policy 1 = trust the caller, 2 = after the concurrency probe -- which may refuse on a box whose queues are taken):
    a lone caller's call forks from three 1080p frames' worth of pixels up -- not below, not for a caller that pipelines
    (concurrent_batches >= 2), not with the stage timers on, not inside a recording."""
    one = images(1, 1920, 1080)
    four = one * 4
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=8192)
    with capi.Context(0) as c:
        assert c.get_policy(capi.POLICY_SIDE_STREAM) == 0
        run_batch(c, four, prm)
        assert c.forks() == 0  # the default
        c.set_policy(capi.POLICY_SIDE_STREAM, 2)
        cnt_p, pts_p = run_batch(c, four, prm)
        assert c.forks() in (0, 1)  # the probe decides; either way the records are the same
        c.set_policy(capi.POLICY_SIDE_STREAM, 0)
        with pytest.raises(capi.CusiftError):
            c.set_policy(capi.POLICY_SIDE_STREAM, 7)
        with pytest.raises(capi.CusiftError):
            c.set_policy(99, 1)
    with capi.Context(0) as c:
        c.set_policy(capi.POLICY_SIDE_STREAM, 1)
        cnt1, pts1 = run_batch(c, one, prm)
        assert c.forks() == 0
        cnt4, pts4 = run_batch(c, four, prm)
        assert c.forks() == 1
        np.testing.assert_array_equal(cnt4, cnt_p)
        for i in range(4):
            assert same_records(canonical_order(pts4[i, : cnt4[i]]), canonical_order(pts_p[i, : cnt_p[i]]))
        piped = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=8192, concurrent_batches=4)
        run_batch(c, four, piped)
        assert c.forks() == 1
        c.timing_enable(True)
        run_batch(c, four, prm)
        c.timing_enable(False)
        assert c.forks() == 1
        stack = np.stack([pitched(i) for i in four])
        d_imgs = DeviceBuffer.from_numpy(c, stack)
        d_pts = DeviceBuffer(c, 4 * prm.max_pts * 588)
        d_cnt = DeviceBuffer(c, 16)
        g = c.record_graph(d_imgs.ptr, 4, 1920, 1080, stack.shape[2], 1080 * stack.shape[2], prm, d_pts.ptr, d_cnt.ptr)
        assert c.forks() == 1
        g.launch()
        c.synchronize()
        g.close()
        for b in (d_imgs, d_pts, d_cnt):
            b.free()
        ref_cnt, ref_pts = run_batch(one_stream, four, prm)
        np.testing.assert_array_equal(cnt4, ref_cnt)
        for i in range(4):
            assert same_records(canonical_order(pts4[i, : cnt4[i]]), canonical_order(ref_pts[i, : ref_cnt[i]]))
            assert same_records(canonical_order(pts4[i, : cnt4[i]]), canonical_order(pts1[0, : cnt1[0]]))
