"""The optional 160-byte wire format of SiftData (cusift_pack_points_compact / cusift_expand_points_host): header fields
exact, descriptor as 128 bytes with one step per record.  New functionality (the reference copies whole 588-byte
records, cuSIFT.cu:52-59); the checker is the formula of include/cusift_amd.h restated in numpy, byte for byte."""
import numpy as np
import pytest

from cusift_amd import capi
from cusift_amd.capi import COMPACT_POINT_DTYPE, SIFT_POINT_DTYPE


def compact_reference(pts):
    """numpy restatement: step = max(data) / 255 (float32 IEEE), q = min(255, floor(data / step + 0.5))."""
    out = np.zeros(len(pts), dtype=COMPACT_POINT_DTYPE)
    for f in ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling"):
        out[f] = pts[f]
    with np.errstate(invalid="ignore", divide="ignore"):
        m = pts["data"].max(axis=1)  # NaN propagates
        ok = (m > 0) & np.isfinite(m)
        step = np.where(ok, (m / np.float32(255.0)).astype(np.float32), m).astype(np.float32)
        q = np.floor(pts["data"] / step[:, None] + np.float32(0.5)).astype(np.float32)
        q = np.clip(q, 0, 255)
    out["desc_step"] = step
    out["q"] = np.where(ok[:, None], q, 0).astype(np.uint8)
    return out


def test_expand_host_round_trip():
    rng = np.random.default_rng(3)
    c = np.zeros(50, dtype=COMPACT_POINT_DTYPE)
    c["coords2D"] = rng.uniform(0, 2000, (50, 2)).astype(np.float32)
    c["scale"] = rng.uniform(1, 30, 50).astype(np.float32)
    c["orientation"] = rng.uniform(0, 360, 50).astype(np.float32)
    c["subsampling"] = 2.0 ** rng.integers(0, 5, 50)
    c["desc_step"] = rng.uniform(1e-4, 2e-3, 50).astype(np.float32)
    c["q"] = rng.integers(0, 256, (50, 128))
    c["desc_step"][7] = np.nan
    p = capi.expand_points(c)
    assert p.dtype == SIFT_POINT_DTYPE
    for f in ("coords2D", "scale", "orientation", "subsampling"):
        np.testing.assert_array_equal(p[f], c[f])
    want = c["q"].astype(np.float32) * c["desc_step"][:, None]
    np.testing.assert_array_equal(p["data"], want)  # NaN row included (equal_nan positional)
    assert not p["score"].any() and not p["match"].any() and not p["coords3D"].any()


@pytest.mark.gpu
def test_compact_pack_matches_the_formula_and_bounds_the_error(ctx, gray1):
    prm = capi.default_params(num_octaves=4, init_blur=0.0, peak_thresh=1.0, max_pts=4096)
    imgs = np.stack([gray1, np.full_like(gray1, 9.0), gray1[::-1].copy()])  # image 1 has no keypoints
    p = capi.ialign_up(640, 128)
    d_imgs = capi.DeviceBuffer.from_numpy(ctx, imgs)
    d_pts = capi.DeviceBuffer(ctx, 3 * prm.max_pts * 588)
    d_cnt = capi.DeviceBuffer(ctx, 12)
    ctx.extract_batch(d_imgs.ptr, 3, 640, 480, p, 480 * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    cnt = np.minimum(d_cnt.to_numpy(np.uint32, (3,)), prm.max_pts)
    assert cnt[0] > 500 and cnt[1] == 0 and cnt[2] > 500
    rec = d_pts.to_numpy(np.uint8, (3 * prm.max_pts, 588)).view(SIFT_POINT_DTYPE).reshape(3, prm.max_pts)
    # plant the special cases in image 0: a NaN descriptor (flat patch), an all-zero one, one with a large element
    rec[0, 5]["data"][:] = np.nan
    rec[0, 6]["data"][:] = 0.0
    rec[0, 7]["data"][3] = 1000.0
    ctx.h2d(d_pts.ptr, rec.view(np.uint8).reshape(-1))
    total = int(cnt.sum())
    d_out = capi.DeviceBuffer(ctx, (total + 8) * 160)
    ctx.memset(d_out.ptr, 0xEE, d_out.nbytes)
    d_off = capi.DeviceBuffer(ctx, 16)
    ctx.pack_points_compact(d_pts.ptr, d_cnt.ptr, 3, prm.max_pts, d_out.ptr, total, d_off.ptr)
    ctx.synchronize()
    off = d_off.to_numpy(np.uint32, (4,))
    np.testing.assert_array_equal(off, [0, cnt[0], cnt[0], total])
    raw = d_out.to_numpy(np.uint8, (total + 8, 160))
    assert (raw[total:] == 0xEE).all()  # nothing beyond `capacity`
    got = raw[:total].copy().view(COMPACT_POINT_DTYPE).reshape(-1)
    flat = np.concatenate([rec[i, : cnt[i]] for i in range(3)])
    want = compact_reference(flat)
    assert got.tobytes() == want.tobytes()
    # what the wire format costs: per-element error <= step / 2, L2 of a whole descriptor a few 1e-3
    back = capi.expand_points(got)
    fin = np.isfinite(flat["data"]).all(axis=1) & (flat["data"].max(axis=1) < 10)
    err = np.abs(back["data"][fin] - flat["data"][fin])
    assert (err <= got["desc_step"][fin][:, None] * 0.5001 + 1e-7).all()  # half a step (+ the rounding of the division)
    l2 = np.linalg.norm(err.astype(np.float64), axis=1)
    assert l2.max() < 1e-2 and np.median(l2) < 5e-3
    assert np.isnan(back["data"][5]).all() and not back["data"][6].any()


# ---- the trimmed 540-byte record: the 135 floats extraction writes, exact ----
WRITTEN = ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data")
UNWRITTEN = ("score", "ambiguity", "match", "match_xpos", "match_ypos", "match_error", "empty", "coords3D")


def test_expand_trimmed_host_round_trip():
    rng = np.random.default_rng(4)
    t = np.zeros(40, dtype=capi.TRIMMED_POINT_DTYPE)
    for f in ("coords2D", "scale", "sharpness", "edgeness", "orientation", "data"):
        t[f] = rng.normal(size=t[f].shape).astype(np.float32)
    t["subsampling"] = 2.0 ** rng.integers(0, 5, 40)
    t["data"][3] = np.nan
    p = capi.expand_trimmed(t)
    assert p.dtype == SIFT_POINT_DTYPE
    for f in WRITTEN:
        assert np.array_equal(p[f], t[f], equal_nan=True), f
    for f in UNWRITTEN:
        assert not np.ascontiguousarray(p[f]).view(np.uint8).any(), f


@pytest.mark.gpu
def test_trimmed_pack_and_expand_are_exact(ctx, gray1):
    prm = capi.default_params(num_octaves=4, init_blur=0.0, peak_thresh=1.0, max_pts=4096)
    imgs = np.stack([gray1, np.full_like(gray1, 9.0), gray1[::-1].copy()])
    p = capi.ialign_up(640, 128)
    d_imgs = capi.DeviceBuffer.from_numpy(ctx, imgs)
    d_pts = capi.DeviceBuffer(ctx, 3 * prm.max_pts * 588)
    ctx.memset(d_pts.ptr, 0x5A, d_pts.nbytes)  # the fields extraction does not write hold "whatever the buffer held"
    d_cnt = capi.DeviceBuffer(ctx, 12)
    ctx.extract_batch(d_imgs.ptr, 3, 640, 480, p, 480 * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    cnt = np.minimum(d_cnt.to_numpy(np.uint32, (3,)), prm.max_pts)
    rec = d_pts.to_numpy(np.uint8, (3 * prm.max_pts, 588)).view(SIFT_POINT_DTYPE).reshape(3, prm.max_pts)
    flat = np.concatenate([rec[i, : cnt[i]] for i in range(3)])
    total = int(cnt.sum())
    assert total > 1000
    d_out = capi.DeviceBuffer(ctx, (total + 8) * 540)
    ctx.memset(d_out.ptr, 0xEE, d_out.nbytes)
    d_off = capi.DeviceBuffer(ctx, 16)
    ctx.pack_points_trimmed(d_pts.ptr, d_cnt.ptr, 3, prm.max_pts, d_out.ptr, total, d_off.ptr)
    ctx.synchronize()
    np.testing.assert_array_equal(d_off.to_numpy(np.uint32, (4,)), [0, cnt[0], cnt[0], total])
    raw = d_out.to_numpy(np.uint8, (total + 8, 540))
    assert (raw[total:] == 0xEE).all()  # nothing beyond `capacity`
    got = raw[:total].copy().view(capi.TRIMMED_POINT_DTYPE).reshape(-1)
    for f in WRITTEN:  # bit for bit, NaN descriptors included
        assert np.ascontiguousarray(got[f]).tobytes() == np.ascontiguousarray(flat[f]).tobytes(), f
    # expansion on the device and on the host: the written fields back in place, the others zero
    d_back = capi.DeviceBuffer(ctx, (total + 1) * 588)
    ctx.memset(d_back.ptr, 0x77, d_back.nbytes)
    ctx.expand_trimmed(d_out.ptr, total, d_back.ptr)
    ctx.synchronize()
    back_raw = d_back.to_numpy(np.uint8, (total + 1, 588))
    assert (back_raw[total] == 0x77).all()
    back = back_raw[:total].copy().view(SIFT_POINT_DTYPE).reshape(-1)
    assert back.tobytes() == capi.expand_trimmed(got).tobytes()
    for f in WRITTEN:
        assert np.ascontiguousarray(back[f]).tobytes() == np.ascontiguousarray(flat[f]).tobytes(), f
    for f in UNWRITTEN:
        assert not np.ascontiguousarray(back[f]).view(np.uint8).any(), f
    for b in (d_imgs, d_pts, d_cnt, d_out, d_off, d_back):
        b.free()
