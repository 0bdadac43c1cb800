"""cusift_pipe_*: the host-to-host pipeline behind the C ABI (frames in host memory in, SiftData in pinned host memory
out, several batches in flight).  Every batch must come back exactly as a blocking cusift_extract_batch of the same
frames would leave it: same counts, same records as sets (the append order inside an octave is racy, as in the
reference), image by image, whatever is in flight beside it."""
import numpy as np
import pytest

from cusift_amd import capi, synth
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer
from oracle_binding import pitched
from parity_utils import canonical_order

pytestmark = pytest.mark.gpu

FIELDS = ("coords2D", "scale", "sharpness", "edgeness", "orientation", "subsampling", "data")


def reference_batch(ctx, frames_f32, prm):
    """blocking extraction of [n, h, w] float frames through cusift_extract_batch; list of per-image record arrays"""
    n, h, w = frames_f32.shape
    p = capi.ialign_up(w, 128)
    stack = np.stack([pitched(f) for f in frames_f32]) if p != w else frames_f32
    d_imgs = DeviceBuffer.from_numpy(ctx, stack)
    d_pts = DeviceBuffer(ctx, n * prm.max_pts * 588)
    d_cnt = DeviceBuffer(ctx, 4 * n)
    ctx.extract_batch(d_imgs.ptr, n, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    cnt = np.minimum(d_cnt.to_numpy(np.uint32, (n,)), prm.max_pts)
    pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (n, prm.max_pts))
    out = [pts[i, : cnt[i]].copy() for i in range(n)]
    for b in (d_imgs, d_pts, d_cnt):
        b.free()
    return out


def same_set(a, b):
    a, b = canonical_order(a), canonical_order(b)
    return len(a) == len(b) and all(np.array_equal(a[f], b[f], equal_nan=True) for f in FIELDS)


@pytest.mark.parametrize("fmt,w,h", [("u8", 640, 480), ("f32", 640, 480), ("u8", 333, 200), ("f32", 333, 200)])
def test_pipe_equals_blocking_extraction(ctx, gray1, fmt, w, h):
    prm = capi.default_params(num_octaves=4, init_blur=0.0, peak_thresh=2.0, max_pts=4096)
    rng = np.random.default_rng(11)
    base = gray1[:h, :w] if (h, w) != gray1.shape else gray1

    def batch(k, n):  # n different 8-bit-valued frames
        fr = [np.roll(base, (7 * k + 3 * i, 11 * k + 5 * i), axis=(0, 1)) for i in range(n)]
        fr[n // 2] = np.clip(fr[n // 2] + rng.integers(-20, 20, size=base.shape), 0, 255).astype(np.float32)
        return np.ascontiguousarray(np.stack(fr), dtype=np.float32)

    sizes = [3, 3, 1, 3, 2, 3, 3]  # batches smaller than n_images are allowed
    frames = [batch(k, n) for k, n in enumerate(sizes)]
    want = [reference_batch(ctx, f, prm) for f in frames]
    assert sum(len(x) for x in want[0]) > 100
    host = [f.astype(np.uint8) if fmt == "u8" else f for f in frames]
    assert all(np.array_equal(hf.astype(np.float32), f) for hf, f in zip(host, frames))
    depth = 3
    with capi.Pipe(0, 3, w, h, prm, capi.PIPE_U8 if fmt == "u8" else capi.PIPE_F32, depth=depth) as pipe:
        with pytest.raises(capi.CusiftError, match="nothing in flight"):
            pipe.collect()
        done = 0
        for k in range(len(frames)):
            if pipe.in_flight() == depth:  # keep the pipeline full: collect the oldest only when there is no room
                with pytest.raises(capi.CusiftError, match="in flight already"):
                    pipe.submit(host[k])
                rec, off = pipe.collect()
                check(rec, off, want[done])
                done += 1
            pipe.submit(host[k])
        while pipe.in_flight():
            rec, off = pipe.collect()
            check(rec, off, want[done])
            done += 1
        assert done == len(frames)
        with pytest.raises(capi.CusiftError, match="images per batch"):
            pipe.submit(np.zeros((4, h, w), np.uint8 if fmt == "u8" else np.float32))


def check(rec, off, want):
    assert len(off) == len(want) + 1 and off[0] == 0 and off[-1] == len(rec)
    for i, w_i in enumerate(want):
        got = rec[off[i]: off[i + 1]]
        assert np.all(np.diff(got["subsampling"]) <= 0)  # coarsest octave first inside every image
        assert same_set(got, w_i), i


def test_pipe_collected_view_survives_submits_at_full_depth(ctx, gray1):
    """The contract of cusift_pipe_collect's pointers: valid until the NEXT collect, through any number of submits in
    between.  At full depth the first submit after a collect re-uses the DEVICE slot just collected -- the pinned host
    buffers are one more than the slots, so the view handed out must not change under it (round 4's header said "until
    depth further submits", which the slot-indexed buffers did not hold: the advisor's finding)."""
    prm = capi.default_params(num_octaves=4, init_blur=0.0, peak_thresh=1.0, max_pts=4096)
    frames = [np.ascontiguousarray(np.stack([np.roll(gray1, (13 * k + i, 29 * k + 3 * i), axis=(0, 1)) for i in range(2)]),
                                   dtype=np.uint8) for k in range(7)]
    depth = 2
    with capi.Pipe(0, 2, 640, 480, prm, capi.PIPE_U8, depth=depth) as pipe:
        for k in range(depth):
            pipe.submit(frames[k])
        for k in range(depth, len(frames)):
            rec, off = pipe.collect()  # views, not copies
            snap_rec, snap_off = rec.copy(), off.copy()
            assert len(rec) > 500
            pipe.submit(frames[k])  # full again: re-uses the device slot of the batch just collected
            # let the new batch's offsets copy and records copy land before looking (they are what would tear the view)
            import time
            time.sleep(0.05)
            assert np.array_equal(off, snap_off) and rec.tobytes() == snap_rec.tobytes()
        while pipe.in_flight():
            pipe.collect()


def test_pipe_capacity_overflow_is_an_error(ctx, gray1):
    prm = capi.default_params(num_octaves=3, init_blur=0.0, peak_thresh=0.5, max_pts=4096)
    frames = np.stack([gray1, gray1[::-1].copy()]).astype(np.uint8)
    with capi.Pipe(0, 2, 640, 480, prm, capi.PIPE_U8, depth=2, records_capacity=100) as pipe:
        pipe.submit(frames)
        with pytest.raises(capi.CusiftError, match="sized for"):
            pipe.collect()
        with pytest.raises(capi.CusiftError, match="destroy"):
            pipe.submit(frames)
    with pytest.raises(capi.CusiftError):
        capi.Pipe(0, 2, 640, 480, prm, 7)
    with pytest.raises(capi.CusiftError):
        capi.Pipe(0, 2, 640, 480, prm, capi.PIPE_U8, depth=1)


def test_pipe_1080p_batches_in_flight(ctx):
    """BASELINE configs[2]'s shape in small: 8 x 1080p per batch, six batches through a pipeline four deep."""
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=8192)
    w, h, n = 1920, 1080, 8
    batches = [np.stack([synth.tile(3000 + 8 * k + i, w, h, 1.0) for i in range(n)]) for k in range(2)]
    want = [reference_batch(ctx, b, prm) for b in batches]
    host = [b.astype(np.uint8) for b in batches]
    assert all(np.array_equal(a.astype(np.float32), b) for a, b in zip(host, batches))  # the generator rounds to integers
    with capi.Pipe(0, n, w, h, prm, capi.PIPE_U8, depth=4, records_capacity=n * 4096) as pipe:
        order = [0, 1, 0, 0, 1, 1]
        out = []
        for k, b in enumerate(order):
            if pipe.in_flight() == 4:
                out.append(tuple(x.copy() for x in pipe.collect()))
            pipe.submit(host[b])
        while pipe.in_flight():
            out.append(tuple(x.copy() for x in pipe.collect()))
        assert len(out) == len(order)
        for (rec, off), b in zip(out, order):
            check(rec, off, want[b])
