"""GPU parity: every HIP stage, called through the C ABI, against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): keypoint location/scale/orientation within 1e-3 (octave pixels / sigma /
degrees); descriptors within 1e-4 L2 -- for EVERY keypoint, no tolerated fraction.  What is actually demanded
is stricter: the filter stages, locations, scales, sharpness, edgeness and orientations are bit-identical
(same fmaf chains, same written-out transcendental functions: cusift_amd/csrc/sift_math.h is compiled into the
kernels and into the oracle); descriptors differ only by the summation order of the histogram (the kernel
gathers per cell, the oracle walks the samples) and must stay below 1e-4 L2 for every point.
"""
import ctypes as C
import os

import numpy as np
import pytest

from cusift_amd import capi
from cusift_amd import synth
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer
from oracle_binding import pitched
from parity_utils import golden_gates, ang_diff, canonical_order, match_nearest, xys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

REF_PARAMS = dict(num_octaves=6, init_blur=0.0, peak_thresh=0.1, edge_thresh=10.0, lowest_scale=0.0,
                  subsampling=1.0, max_pts=16384)


def rand_image(h, w, seed):
    rng = np.random.default_rng(seed)
    # smooth-ish integer image: noise + low-frequency structure
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = 128 + 60 * np.sin(x / 7.0 + seed) * np.cos(y / 5.0) + rng.uniform(-40, 40, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# The written-out transcendental functions: device == host, bit for bit
# ------------------------------------------------------------------------------------------------
def test_math_device_equals_host(ctx, oracle):
    """cusift_amd/csrc/sift_math.h is compiled into the kernels and into the oracle; the device build must give the
    host build's bits for every input -- normal, denormal, zero, infinite, NaN (cusift_math_eval)."""
    rng = np.random.default_rng(5)
    n = 1 << 20
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-39, 1.17549435e-38, 3.4e38, -3.4e38,
                        1.0, -1.0, 0.5, 88.7, 88.8, -87.4, -103.9, -104.1, 127.9, 128.0, -126.5, -149.5, -150.5],
                       dtype=np.float32)

    def bits(a):
        return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

    def same(a, b):
        nan = np.isnan(a) & np.isnan(b)
        return bool(np.all((bits(a) == bits(b)) | nan))

    def inputs(kind):
        if kind == "exp":
            x = np.concatenate([rng.uniform(-110, 92, n), rng.uniform(-1, 1, n)])
        elif kind == "exp2":
            x = np.concatenate([rng.uniform(-155, 130, n), rng.uniform(-1, 1, n)])
        elif kind == "sincos":
            x = np.concatenate([rng.uniform(0, 6.3, n), rng.uniform(-2e9, 2e9, n)])
        else:
            x = rng.normal(0, 1, 2 * n) * np.exp(rng.uniform(-90, 90, 2 * n))
        with np.errstate(over="ignore"):
            x = x.astype(np.float32)
        raw = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)  # any bit pattern
        return np.concatenate([x, raw, special])

    for op, kind in enumerate(("exp", "exp2", "atan2", "sincos")):
        a = inputs(kind)
        b = inputs(kind) if kind == "atan2" else a
        if kind == "atan2":  # every pair of special values as well
            Y, X = [v.ravel() for v in np.meshgrid(special, special, indexing="ij")]
            a, b = np.concatenate([a, Y]), np.concatenate([b, X])
        d_a, d_b = DeviceBuffer.from_numpy(ctx, a), DeviceBuffer.from_numpy(ctx, b)
        d_o, d_o2 = DeviceBuffer(ctx, a.nbytes), DeviceBuffer(ctx, a.nbytes)
        ctx.math_eval(op, d_a.ptr, d_b.ptr, d_o.ptr, d_o2.ptr, a.size)
        ctx.synchronize()
        got = d_o.to_numpy(np.float32, a.shape)
        want = oracle.math_eval(kind, a, b)
        if kind == "sincos":
            assert same(got, want[0]) and same(d_o2.to_numpy(np.float32, a.shape), want[1]), kind
        else:
            bad = np.nonzero(~((bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))))[0]
            assert bad.size == 0, (kind, a[bad[:5]], b[bad[:5]], got[bad[:5]], want[bad[:5]])
        for buf in (d_a, d_b, d_o, d_o2):
            buf.free()


def test_orientation_bin_shortcut(ctx, oracle):
    """The orientation stage's histogram bin without an angle (sift_keypoints.hip: ori_bin_shortcut, cusift_math_eval op 5):
    edges passed inside the octant instead of (int)(16 atan2f / 3.1416f + 16.5f).  Every sample the shortcut does NOT report
    `near` an edge must get exactly the reference formula's bin (the kernel sends waves with a `near` sample through the
    formula itself): random directions over twelve binades, directions dense at each of the 32 edges, integer-valued
    gradients as an image's are, zeros, signed zeros, equal magnitudes; non-finite gradients must be `near`."""
    rng = np.random.default_rng(23)
    n = 1 << 21
    dy = (rng.normal(0, 1, n) * np.exp(rng.uniform(-6, 6, n))).astype(np.float32)
    dx = (rng.normal(0, 1, n) * np.exp(rng.uniform(-6, 6, n))).astype(np.float32)
    # dense at the edges: theta_m = (m - 16.5) 3.1416 / 16, m = 1 .. 32, +- 2e-4 rad (the `near` margin is ~1.5e-5 rad)
    m = rng.integers(1, 33, n)
    th = (m - 16.5) * 3.1416 / 16.0 + rng.uniform(-2e-4, 2e-4, n)
    r = np.exp(rng.uniform(-4, 6, n))
    ey, ex = (r * np.sin(th)).astype(np.float32), (r * np.cos(th)).astype(np.float32)
    # what a texture tap difference of an 8-bit image looks like: multiples of 1/256 up to +-255
    iy = (rng.integers(-65280, 65281, n) / 256.0).astype(np.float32)
    ix = (rng.integers(-65280, 65281, n) / 256.0).astype(np.float32)
    sy = np.float32([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 1.0, -1.0, 0.0, -0.0, 3.0, -3.0, 2.5, 2.5, -2.5, -2.5])
    sx = np.float32([0.0, 0.0, -0.0, -0.0, 0.0, 0.0, -0.0, -0.0, 2.0, -2.0, 0.0, 0.0, 2.5, -2.5, 2.5, -2.5])
    bad_y = np.float32([np.nan, 1.0, np.inf, -np.inf, 1.0, np.nan])
    bad_x = np.float32([1.0, np.nan, 1.0, np.inf, np.inf, np.nan])
    a = np.concatenate([dy, ey, iy, sy, bad_y])
    b = np.concatenate([dx, ex, ix, sx, bad_x])
    d_a, d_b = DeviceBuffer.from_numpy(ctx, a), DeviceBuffer.from_numpy(ctx, b)
    d_o, d_o2 = DeviceBuffer(ctx, a.nbytes), DeviceBuffer(ctx, a.nbytes)
    ctx.math_eval(5, d_a.ptr, d_b.ptr, d_o.ptr, d_o2.ptr, a.size)
    ctx.synchronize()
    got = d_o.to_numpy(np.float32, a.shape).astype(np.int32)
    ref_dev = d_o2.to_numpy(np.float32, a.shape).astype(np.int32)
    for buf in (d_a, d_b, d_o, d_o2):
        buf.free()
    near, bins = got >= 64, got & 63
    # the reference formula as the ORACLE evaluates it (sm_atan2f on the host == on the device, bit for bit)
    theta = oracle.math_eval("atan2", a, b)
    with np.errstate(invalid="ignore"):
        u = np.float32(16.0) * theta / np.float32(3.1416) + np.float32(16.5)
        want = np.where(np.isfinite(u), u, 0).astype(np.int32)
    want[(want > 31) | (want < 0)] = 0
    fin = np.isfinite(a) & np.isfinite(b)
    np.testing.assert_array_equal(ref_dev[fin], want[fin])  # the device's own evaluation of the formula
    clear = ~near
    assert (bins[clear] == want[clear]).all(), int((bins[clear] != want[clear]).sum())
    assert (bins >= 0).all() and (bins <= 31).all()
    # how often the kernel pays for the formula: per sample, on random directions / on an image's kind of gradients
    assert near[:n].mean() < 6e-4 and near[2 * n:3 * n].mean() < 6e-4, (near[:n].mean(), near[2 * n:3 * n].mean())
    assert 0.03 < near[n:2 * n].mean() < 0.3  # the dense-at-the-edges sample does hit the margin
    # zero gradients carry weight 0 and never ask for the formula; non-finite ones always do
    zero = (a == 0) & (b == 0)
    assert zero.sum() >= 4 and not near[zero].any()
    assert near[~fin].all()


def test_descriptor_angle_coordinate(ctx, oracle):
    """The descriptor stage's angle coordinate (sift_keypoints.hip: desc_angle_bins, cusift_math_eval op 4) is NOT the
    oracle's 4/3.1415f * atan2f + 4 to the ulp -- its effect on a descriptor is continuous, so it is a degree-4 fit -- except
    where it decides: at 8.0 the reference's index becomes 8 and the share lands in the next cell (cuSIFT_D.cu:233-255).
    ONE bound, stated the same way in sift_keypoints.hip (desc_angle_bins) and include/cusift_amd_stages.h: Q of degree 4 in
    s = t^2, fit error 3.1e-6 of a bin, <= 4e-6 of a bin on the device with v_rcp_f32's last ulp (measured 3.8e-6).
    So: within 4e-6 of the oracle's value everywhere, and the SAME side of 8.0 (and of every other integer that close to
    the negative x axis) for every input -- with the operands that sit on the jump sampled densely."""
    rng = np.random.default_rng(11)
    n = 1 << 20
    dy = (rng.normal(0, 1, n) * np.exp(rng.uniform(-6, 6, n))).astype(np.float32)
    dx = (rng.normal(0, 1, n) * np.exp(rng.uniform(-6, 6, n))).astype(np.float32)
    # next to the negative x axis, both signs of dy: |dy| / |dx| from 0 to 3e-4 (the jump sits at 9.27e-5) ...
    ax = np.exp(rng.uniform(-5, 5, n)).astype(np.float32)
    t = rng.uniform(0.0, 3.0e-4, n)
    near_dy = (t * ax * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    # ... and the jump itself, densely
    t2 = rng.uniform(9.2e-5, 9.35e-5, n)
    jump_dy = (t2 * ax).astype(np.float32)
    axis = np.array([0.0, -0.0, 1e-30, -1e-30], dtype=np.float32)
    a = np.concatenate([dy, near_dy, jump_dy, np.repeat(axis, 4), np.float32([1.0, -1.0, 1.0, -1.0])])
    b = np.concatenate([dx, -ax, -ax, np.tile(np.float32([-1.0, -3.5, -100.0, -1e-3]), 4), np.float32([1.0, 1.0, -1.0, -1.0])])
    d_a, d_b = DeviceBuffer.from_numpy(ctx, a), DeviceBuffer.from_numpy(ctx, b)
    d_o = DeviceBuffer(ctx, a.nbytes)
    ctx.math_eval(4, d_a.ptr, d_b.ptr, d_o.ptr, None, a.size)
    ctx.synchronize()
    got = d_o.to_numpy(np.float32, a.shape)
    th = oracle.math_eval("atan2", a, b)  # sm_atan2f, the oracle's and the orientation stage's
    want = (np.float32(4.0) / np.float32(3.1415)) * th + np.float32(4.0)  # float32 throughout, as cuSIFT_D.cu:233 / the oracle
    assert want.dtype == np.float32
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert err.max() < 4e-6, (err.max(), a[err.argmax()], b[err.argmax()])
    near = slice(n, a.size - 4)  # everything next to the negative x axis (elsewhere an integer crossed is a continuous event)
    # (int) truncates, as the kernels' v_cvt_i32_f32.  The quotient |dy| / |dx| is v_rcp_f32's (1 ulp, as it was before the
    # fit): a pair whose exact quotient lies within an ulp (7e-12) of the threshold may land on the other side -- ~5e-6 of
    # a sample as dense as this one (2^20 operands within 1.5e-6 of the jump), ~1e-12 of an image's gradients
    flips = int((got[near].astype(np.int32) != want[near].astype(np.int32)).sum())
    assert flips <= 20, flips
    np.testing.assert_array_equal(got[n:2 * n].astype(np.int32)[np.abs(t - 9.27e-5) > 1e-6],
                                  want[n:2 * n].astype(np.int32)[np.abs(t - 9.27e-5) > 1e-6])
    eight = want >= 8.0
    assert 0.2 < eight[2 * n:3 * n].mean() < 0.8  # the dense sample straddles the jump
    assert (got[near].view(np.uint32) == want[near].view(np.uint32)).mean() > 0.999  # the reference's own operations there
    print("descriptor angle coordinate: largest distance to the oracle's %.2e of a bin; index 8 for %d of %d operands at the "
          "jump, %d of all %d near the axis on the other side of it than the oracle" %
          (err.max(), int(eight[2 * n:3 * n].sum()), n, flips, 2 * n))
    for buf in (d_a, d_b, d_o):
        buf.free()


# ------------------------------------------------------------------------------------------------
# ScaleDown
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h", [(640, 480), (101, 77), (130, 64), (2, 2), (7, 9), (1920, 1080), (256, 17), (4, 4),
                                 (248, 6), (252, 31), (500, 3), (960, 540), (120, 67)])
def test_scale_down_bit_exact(ctx, oracle, gray1, w, h):
    img = gray1 if (w, h) == (640, 480) else rand_image(h, w, w * 131 + h)
    src = pitched(img)
    want = oracle.scale_down(src, w, h)
    ow, oh = w // 2, h // 2
    op = want.shape[1]
    d_src = DeviceBuffer.from_numpy(ctx, src)
    d_dst = DeviceBuffer(ctx, max(oh, 1) * op * 4)
    d_dst.zero()
    ctx.scale_down(d_dst.ptr, op, d_src.ptr, w, h, src.shape[1])
    got = d_dst.to_numpy(np.float32, (max(oh, 1), op))
    np.testing.assert_array_equal(got[:oh, :ow], want[:oh, :ow])
    # bounds-checked writes: nothing outside the (w/2) x (h/2) result (the reference overruns)
    assert not got[:oh, ow:].any()


def test_scale_down_batch_strided(ctx, oracle):
    n, w, h = 3, 200, 90
    imgs = np.stack([pitched(rand_image(h, w, 7 + i)) for i in range(n)])
    p = imgs.shape[2]
    op = 128
    d_src = DeviceBuffer.from_numpy(ctx, imgs)
    d_dst = DeviceBuffer(ctx, n * (h // 2) * op * 4)
    d_dst.zero()
    ctx.scale_down(d_dst.ptr, op, d_src.ptr, w, h, p, n_images=n)
    got = d_dst.to_numpy(np.float32, (n, h // 2, op))
    for i in range(n):
        np.testing.assert_array_equal(got[i, :, : w // 2], oracle.scale_down(imgs[i], w, h)[:, : w // 2])


@pytest.mark.parametrize("w,h,n_levels,n", [(1920, 1080, 4, 1), (1027, 301, 4, 2), (333, 257, 4, 3), (640, 480, 3, 1),
                                           (131, 200, 2, 2), (64, 48, 1, 1), (37, 61, 4, 1), (16, 16, 4, 1),
                                           (1366, 768, 4, 1), (19, 500, 4, 2)])
def test_scale_down_levels_equals_the_chain(ctx, oracle, w, h, n_levels, n):
    """cusift_scale_down_levels (the ScaleDown chain of a small call in ONE launch: every workgroup recomputes in LDS
    what it needs of every level) == the chain of ScaleDowns, bit for bit at every level -- odd sizes, images narrower
    than a workgroup's needed square, levels that shrink to one pixel, batches."""
    imgs = np.stack([pitched(rand_image(h, w, 50 + i)) for i in range(n)])
    p = imgs.shape[2]
    d_src = DeviceBuffer.from_numpy(ctx, imgs)
    dims = [(w >> k, h >> k) for k in range(1, n_levels + 1)]
    assert all(a >= 1 and b >= 1 for a, b in dims)
    pitches = [capi.ialign_up(a, 128) for a, _ in dims]
    bufs = [DeviceBuffer(ctx, n * b * pk * 4) for (a, b), pk in zip(dims, pitches)]
    for b in bufs:
        b.zero()
    ctx.scale_down_levels(d_src.ptr, w, h, p, [b.ptr for b in bufs], pitches, n_images=n)
    ctx.synchronize()
    for i in range(n):
        prev, pw, ph = imgs[i], w, h
        for k, ((a, b), pk) in enumerate(zip(dims, pitches)):
            want = oracle.scale_down(prev, pw, ph)  # [ph // 2, >= pw // 2]
            got = bufs[k].to_numpy(np.float32, (n, b, pk))[i]
            np.testing.assert_array_equal(got[:, :a], want[:b, :a], err_msg="image %d level %d" % (i, k + 1))
            assert not got[:, a:].any()  # nothing written beyond the level's columns
            prev = np.zeros((b, capi.ialign_up(a, 128)), dtype=np.float32)
            prev[:, :a] = want[:b, :a]
            pw, ph = a, b
    with pytest.raises(capi.CusiftError):
        ctx.scale_down_levels(d_src.ptr, w, h, p, [bufs[0].ptr] * 5, [pitches[0]] * 5, n_images=n)  # at most 4 levels
    for b in bufs + [d_src]:
        b.free()


# ------------------------------------------------------------------------------------------------
# LaplaceMulti (blur + DoG)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize(
    "w,h,blur",
    [(640, 480, 0.0), (640, 480, 0.5590170), (320, 240, 1.0), (250, 40, 0.0), (249, 33, 0.0), (5, 3, 0.0),
     (1, 1, 0.0), (131, 200, 0.3), (500, 9, 0.0), (1920, 1080, 0.0)],
)
def test_laplace_multi_bit_exact(ctx, oracle, gray1, w, h, blur):
    if (w, h) == (640, 480):
        img = gray1
    elif (w, h) == (320, 240):
        img = gray1[::2, ::2].copy()
    else:
        img = rand_image(h, w, w * 7 + h)
    src = pitched(img)
    p = src.shape[1]
    want = oracle.laplace_multi(src, w, h, blur)
    d_src = DeviceBuffer.from_numpy(ctx, src)
    d_dog = DeviceBuffer(ctx, 7 * h * p * 4)
    d_dog.zero()
    ctx.laplace_multi(d_src.ptr, w, h, p, blur, d_dog.ptr)
    got = d_dog.to_numpy(np.float32, (7, h, p))
    assert np.isfinite(got).all()
    np.testing.assert_array_equal(got[:, :, :w], want[:, :, :w])
    assert not got[:, :, w:].any()  # pad columns are not written (cuSIFT_D.cu:551)


def test_laplace_multi_unaligned_pitch_uses_scalar_path(ctx, oracle):
    """A caller-owned cuImage may have any pitch (cuImage.cu:16): odd pitch -> scalar loads/stores."""
    w, h, p = 77, 31, 79
    img = rand_image(h, w, 5)
    src = np.zeros((h, p), dtype=np.float32)
    src[:, :w] = img
    want = oracle.laplace_multi(src, w, h, 0.0)
    d_src = DeviceBuffer.from_numpy(ctx, src)
    d_dog = DeviceBuffer(ctx, 7 * h * p * 4)
    d_dog.zero()
    ctx.laplace_multi(d_src.ptr, w, h, p, 0.0, d_dog.ptr)
    got = d_dog.to_numpy(np.float32, (7, h, p))
    np.testing.assert_array_equal(got[:, :, :w], want[:, :, :w])


def test_laplace_multi_linearity(ctx):
    """Full-size property: DoG(a*I + b) == a*DoG(I) up to rounding (taps sum to 1, DoG kills constants)."""
    w, h = 1920, 1080
    img = synth.tile(1000, w, h)
    src = pitched(img)
    p = src.shape[1]
    d_a = DeviceBuffer.from_numpy(ctx, src)
    d_b = DeviceBuffer.from_numpy(ctx, pitched(0.5 * img + 16.0))
    dog_a = DeviceBuffer(ctx, 7 * h * p * 4)
    dog_b = DeviceBuffer(ctx, 7 * h * p * 4)
    ctx.laplace_multi(d_a.ptr, w, h, p, 0.0, dog_a.ptr)
    ctx.laplace_multi(d_b.ptr, w, h, p, 0.0, dog_b.ptr)
    a = dog_a.to_numpy(np.float32, (7, h, p))[:, :, :w]
    b = dog_b.to_numpy(np.float32, (7, h, p))[:, :, :w]
    np.testing.assert_allclose(b, 0.5 * a, atol=2e-4, rtol=0)


# ------------------------------------------------------------------------------------------------
# FindPointsMulti
# ------------------------------------------------------------------------------------------------
def run_find_points(ctx, dog, w, h, thresh, edge, sub, max_pts):
    p = dog.shape[2]
    d_dog = DeviceBuffer.from_numpy(ctx, dog)
    d_pts = DeviceBuffer(ctx, max_pts * 588)
    d_pts.zero()
    d_cnt = DeviceBuffer(ctx, 4)
    d_cnt.zero()
    ctx.find_points_multi(d_dog.ptr, w, h, p, thresh, edge, sub, d_pts.ptr, max_pts, d_cnt.ptr)
    cnt = int(d_cnt.to_numpy(np.uint32, (1,))[0])
    pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (max_pts,))
    return pts, cnt


@pytest.mark.parametrize("w,h,blur,thresh", [(640, 480, 0.0, 0.1), (640, 480, 0.0, 1.0), (320, 240, 0.559, 0.1),
                                             (125, 70, 0.0, 0.5), (3, 3, 0.0, 0.01), (124, 9, 0.0, 0.2)])
def test_find_points_same_set_as_oracle(ctx, oracle, gray1, w, h, blur, thresh):
    if (w, h) == (640, 480):
        img = gray1
    elif (w, h) == (320, 240):
        img = gray1[::2, ::2].copy()
    else:
        img = rand_image(h, w, w + 3 * h)
    src = pitched(img)
    dog = oracle.laplace_multi(src, w, h, blur)
    max_pts = 16384
    want, n_want = oracle.find_points_multi(dog, w, h, thresh, 10.0, 2.0, max_pts)
    got, n_got = run_find_points(ctx, dog, w, h, thresh, 10.0, 2.0, max_pts)
    assert n_got == n_want
    if (w, h) == (640, 480) and thresh == 0.1:
        assert n_got == 7953  # octave 0 of the golden run
    a = canonical_order(want[:n_want])
    b = canonical_order(got[:n_got])
    # location, sharpness, edgeness use IEEE +,-,*,/ only -> bit-exact; scale goes through exp2f
    np.testing.assert_array_equal(a["coords2D"], b["coords2D"])
    np.testing.assert_array_equal(a["sharpness"], b["sharpness"])
    np.testing.assert_array_equal(a["edgeness"], b["edgeness"])
    np.testing.assert_array_equal(a["subsampling"], b["subsampling"])
    np.testing.assert_array_equal(a["scale"], b["scale"])  # exp2f is the shared written-out function


def test_find_points_overflow_is_dropped_not_written(ctx, oracle, gray1):
    w, h = 640, 480
    dog = oracle.laplace_multi(pitched(gray1), w, h, 0.0)
    p = dog.shape[2]
    max_pts = 64
    d_dog = DeviceBuffer.from_numpy(ctx, dog)
    d_pts = DeviceBuffer(ctx, (max_pts + 8) * 588)
    d_pts.zero()
    d_cnt = DeviceBuffer(ctx, 4)
    d_cnt.zero()
    ctx.find_points_multi(d_dog.ptr, w, h, p, 0.1, 10.0, 1.0, d_pts.ptr, max_pts, d_cnt.ptr)
    cnt = int(d_cnt.to_numpy(np.uint32, (1,))[0])
    pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (max_pts + 8,))
    assert cnt == 7953  # the counter keeps counting (cuSIFT_D.cu:513)
    assert (pts["subsampling"][:max_pts] == 1.0).all()
    assert not np.frombuffer(pts[max_pts:].tobytes(), dtype=np.uint8).any()  # guard slots untouched


# ------------------------------------------------------------------------------------------------
# Fused detection (LaplaceMulti + FindPointsMulti in one kernel, DoG never stored)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("w,h,blur,thresh", [(640, 480, 0.0, 0.1), (320, 240, 0.559017, 0.1), (244, 37, 0.0, 0.5),
                                             (1920, 1080, 1.0, 3.0), (8, 5, 0.0, 0.01), (4, 3, 0.0, 0.01),
                                             (480, 9, 0.3, 0.2),
                                             # ragged widths (w % 4 != 0): the last column group is partial
                                             (1366, 768, 0.0, 1.0), (249, 37, 0.0, 0.5), (30, 16, 0.0, 0.2),
                                             (15, 8, 0.0, 0.05), (6, 5, 0.0, 0.01), (5, 3, 0.0, 0.01),
                                             (1001, 75, 0.5, 0.5), (243, 9, 0.0, 0.2), (241, 12, 0.0, 0.2)])
def test_detect_multi_same_set_as_oracle(ctx, oracle, gray1, w, h, blur, thresh):
    if (w, h) == (640, 480):
        img = gray1
    elif (w, h) == (320, 240):
        img = gray1[::2, ::2].copy()
    elif (w, h) == (1920, 1080):
        img = synth.tile(1003, preblur=1.0)
    else:
        img = rand_image(h, w, w + 5 * h)
    src = pitched(img)
    dog = oracle.laplace_multi(src, w, h, blur)
    max_pts = 32768
    want, n_want = oracle.find_points_multi(dog, w, h, thresh, 10.0, 4.0, max_pts)
    d_img = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer(ctx, max_pts * 588)
    d_pts.zero()
    d_cnt = DeviceBuffer(ctx, 4)
    d_cnt.zero()
    ctx.detect_multi(d_img.ptr, w, h, src.shape[1], blur, thresh, 10.0, 4.0, d_pts.ptr, max_pts, d_cnt.ptr)
    n_got = int(d_cnt.to_numpy(np.uint32, (1,))[0])
    got = d_pts.to_numpy(SIFT_POINT_DTYPE, (max_pts,))
    assert n_got == n_want
    a, b = canonical_order(want[:n_want]), canonical_order(got[:n_got])
    np.testing.assert_array_equal(a["coords2D"], b["coords2D"])
    np.testing.assert_array_equal(a["sharpness"], b["sharpness"])
    np.testing.assert_array_equal(a["edgeness"], b["edgeness"])
    np.testing.assert_array_equal(a["scale"], b["scale"])


@pytest.mark.parametrize("w,h,blur,thresh,n", [(640, 480, 0.0, 0.5, 1), (1920, 1080, 1.0, 3.0, 1), (101, 77, 0.0, 0.2, 2),
                                               (130, 64, 0.5, 0.2, 1), (7, 9, 0.0, 0.01, 1), (256, 17, 0.0, 0.2, 3),
                                               (4, 4, 0.0, 0.01, 1), (248, 6, 0.0, 0.2, 1), (252, 31, 0.0, 0.2, 1),
                                               (500, 3, 0.0, 0.2, 2), (960, 540, 0.71, 2.0, 1), (120, 67, 0.0, 0.2, 1),
                                               (241, 12, 0.0, 0.2, 1), (243, 9, 0.0, 0.2, 1), (481, 5, 0.0, 0.1, 1),
                                               (1366, 768, 0.0, 1.0, 1), (5, 3, 0.0, 0.01, 1), (239, 200, 1.0, 0.3, 2),
                                               (722, 131, 0.0, 0.3, 1)])
def test_detect_multi_down_emits_scale_down_bit_exact(ctx, oracle, gray1, w, h, blur, thresh, n):
    """The fused detection that also writes the next octave's image (cusift_detect_multi_down: ScaleDown as a by-product
    of the detection's row window, cuSIFT_D.cu:37-182): the image is the oracle's ScaleDown bit for bit -- odd sizes,
    ragged widths, widths around a strip border (240 columns), heights of one chunk and of many, batches, nothing written
    outside (w/2) x (h/2) -- and the keypoint heads are the plain fused detection's records."""
    if (w, h) == (640, 480):
        imgs = [gray1]
    elif (w, h) == (1920, 1080):
        imgs = [synth.tile(1003, preblur=1.0)]
    else:
        imgs = [rand_image(h, w, w * 17 + h + i) for i in range(n)]
    stack = np.stack([pitched(i) for i in imgs])
    p = stack.shape[2]
    ow, oh = w // 2, h // 2
    op = capi.ialign_up(max(ow, 1), 128)
    max_pts = 16384
    d_img = DeviceBuffer.from_numpy(ctx, stack)
    d_next = DeviceBuffer(ctx, n * oh * op * 4)
    ctx.memset(d_next.ptr, 0x5A, n * oh * op * 4)
    d_heads = DeviceBuffer(ctx, n * max_pts * 64)
    d_heads.zero()
    d_cnt = DeviceBuffer(ctx, 4 * n)
    d_cnt.zero()
    ctx.detect_multi_down(d_img.ptr, w, h, p, blur, thresh, 10.0, 2.0, d_heads.ptr, max_pts, d_cnt.ptr, d_next.ptr, op,
                          n_images=n)
    got_next = d_next.to_numpy(np.float32, (n, oh, op))
    cnt = d_cnt.to_numpy(np.uint32, (n,))
    heads = d_heads.to_numpy(np.float32, (n, max_pts, 16))
    # the plain detection of the same images, whole records
    d_pts = DeviceBuffer(ctx, n * max_pts * 588)
    d_pts.zero()
    d_cnt2 = DeviceBuffer(ctx, 4 * n)
    d_cnt2.zero()
    ctx.detect_multi(d_img.ptr, w, h, p, blur, thresh, 10.0, 2.0, d_pts.ptr, max_pts, d_cnt2.ptr, n_images=n)
    cnt2 = d_cnt2.to_numpy(np.uint32, (n,))
    pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (n, max_pts))
    np.testing.assert_array_equal(cnt, cnt2)
    pad = np.frombuffer(b"\x5a" * 4, dtype=np.float32)[0]
    for i in range(n):
        want = oracle.scale_down(stack[i], w, h)
        np.testing.assert_array_equal(got_next[i, :, :ow], want[:oh, :ow])
        assert np.all(got_next[i, :, ow:].view(np.uint32) == pad.view(np.uint32))  # pad columns untouched
        k = int(cnt[i])
        hd = heads[i, :k]
        rec = pts[i, :k]
        a = np.stack([hd[:, 0], hd[:, 1], hd[:, 2], hd[:, 3], hd[:, 4], hd[:, 12]], axis=1)
        b = np.stack([rec["coords2D"][:, 0], rec["coords2D"][:, 1], rec["scale"], rec["sharpness"], rec["edgeness"],
                      rec["subsampling"]], axis=1)
        a = a[np.lexsort(a.T[::-1])]
        b = b[np.lexsort(b.T[::-1])]
        np.testing.assert_array_equal(a, b)
    if (w, h) in ((640, 480), (1920, 1080)):
        assert cnt.sum() > 500
    for buf in (d_img, d_next, d_heads, d_cnt, d_pts, d_cnt2):
        buf.free()


def test_generic_kernels_on_unaligned_pitch(ctx, oracle):
    """A caller-owned cuImage may have any pitch (cuImage.cu:16).  Rows that are not 16-byte aligned take the generic
    kernels (per-column clamps, scalar loads): ScaleDown, LaplaceMulti, FindPointsMulti bit-exact there too."""
    w, h, p = 77, 45, 79
    img = rand_image(h, w, 23)
    src = np.zeros((h, p), dtype=np.float32)
    src[:, :w] = img
    d_src = DeviceBuffer.from_numpy(ctx, src)
    # ScaleDown into an odd-pitched destination
    ow, oh, op = w // 2, h // 2, 39
    d_dst = DeviceBuffer(ctx, oh * op * 4)
    d_dst.zero()
    ctx.scale_down(d_dst.ptr, op, d_src.ptr, w, h, p)
    np.testing.assert_array_equal(d_dst.to_numpy(np.float32, (oh, op))[:, :ow], oracle.scale_down(src, w, h)[:oh, :ow])
    # LaplaceMulti + FindPointsMulti on the odd pitch
    want_dog = oracle.laplace_multi(src, w, h, 0.0)
    d_dog = DeviceBuffer(ctx, 7 * h * p * 4)
    d_dog.zero()
    ctx.laplace_multi(d_src.ptr, w, h, p, 0.0, d_dog.ptr)
    np.testing.assert_array_equal(d_dog.to_numpy(np.float32, (7, h, p))[:, :, :w], want_dog[:, :, :w])
    want, n_want = oracle.find_points_multi(want_dog, w, h, 0.5, 10.0, 1.0, 4096)
    got, n_got = run_find_points(ctx, want_dog, w, h, 0.5, 10.0, 1.0, 4096)
    assert n_got == n_want > 20
    a, b = canonical_order(want[:n_want]), canonical_order(got[:n_got])
    for f in ("coords2D", "scale", "sharpness", "edgeness"):
        np.testing.assert_array_equal(a[f], b[f])
    # the driver on such an image: every octave >= 1 lives in the arena (aligned) and takes the fused kernel, octave 0
    # (odd pitch) the two-stage pair -- one ragged octave does not demote the others
    kw = dict(num_octaves=3, init_blur=0.0, peak_thresh=1.0, max_pts=4096)
    prm = capi.default_params(**kw)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    ctx.timing_enable(True)
    ctx.timing_reset()
    n = ctx.extract(d_src.ptr, w, h, p, prm, d_pts.ptr, h_pts)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    if not os.environ.get("CUSIFT_FORCE_GENERIC"):
        assert t["detect_multi"][1] == 2 and t["laplace_multi"][1] == 1 and t["describe_all"][1] == 1
    compare_sets(oracle.extract(img, **kw), h_pts[:n])


def test_detect_multi_rejects_what_it_cannot_do(ctx):
    d = DeviceBuffer(ctx, 1 << 16)
    with pytest.raises(capi.CusiftError, match="aligned"):
        ctx.detect_multi(d.ptr, 30, 20, 31, 0.0, 1.0, 10.0, 1.0, d.ptr, 16, d.ptr)  # odd pitch, w % 4 != 0


def test_two_stage_and_fused_drivers_agree(ctx, gray1):
    """cusift_params.fused_detect selects the pipeline; both must give the same SiftData (as sets)."""
    res = []
    for fused in (1, 0):
        got = gpu_extract(ctx, gray1, fused_detect=fused, **REF_PARAMS)
        res.append(canonical_order(got))
    assert len(res[0]) == len(res[1]) == 9508
    np.testing.assert_array_equal(res[0]["coords2D"], res[1]["coords2D"])
    np.testing.assert_array_equal(res[0]["orientation"], res[1]["orientation"])
    np.testing.assert_array_equal(res[0]["data"], res[1]["data"])


# ------------------------------------------------------------------------------------------------
# ComputeOrientations / ExtractSiftDescriptors on identical keypoints
# ------------------------------------------------------------------------------------------------
def oracle_octave_points(oracle, img, w, h, blur, thresh, sub, max_pts=16384):
    src = pitched(img)
    dog = oracle.laplace_multi(src, w, h, blur)
    pts, n = oracle.find_points_multi(dog, w, h, thresh, 10.0, sub, max_pts)
    n = min(n, max_pts)
    return src, pts, n


# (8: the reference's texture unit; 0: exact fp32 fractions and products; 4 and 11: the ends of the range in which the
# fixed-point weight rule's product A B / 2^q is exact in fp32 -- the rule is generic in the bit count on both sides)
@pytest.mark.parametrize("frac_bits", [8, 0, 4, 11])
def test_orientations_match_oracle(ctx, oracle, gray1, frac_bits):
    w, h = 640, 480
    src, pts, n = oracle_octave_points(oracle, gray1, w, h, 0.0, 0.5, 1.0)
    assert n > 1000
    want = pts.copy()
    oracle.compute_orientations(src, w, h, want, 0, n, frac_bits)
    d_img = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    d_cnt = DeviceBuffer.from_numpy(ctx, np.array([n], dtype=np.uint32))
    ctx.compute_orientations(d_img.ptr, w, h, src.shape[1], d_pts.ptr, len(pts), None, d_cnt.ptr, frac_bits)
    got = d_pts.to_numpy(SIFT_POINT_DTYPE, (len(pts),))
    # same operations on the same operands (shared expf/atan2f, histogram summed in the oracle's order): same bits,
    # NaN (flat patch, cuSIFT_D.cu:369-383) in the same places
    np.testing.assert_array_equal(got["orientation"][:n], want["orientation"][:n])
    assert np.isfinite(want["orientation"][:n]).mean() > 0.999
    # untouched fields stay untouched
    np.testing.assert_array_equal(want["coords2D"][:n], got["coords2D"][:n])


@pytest.mark.parametrize("frac_bits", [8, 0, 4, 11])
def test_descriptors_match_oracle(ctx, oracle, gray1, frac_bits):
    w, h = 640, 480
    sub = 2.0
    src, pts, n = oracle_octave_points(oracle, gray1, w, h, 0.0, 0.5, sub)
    oracle.compute_orientations(src, w, h, pts, 0, n, frac_bits)
    want = pts.copy()
    oracle.extract_descriptors(src, w, h, want, 0, n, sub, frac_bits)
    d_img = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    d_cnt = DeviceBuffer.from_numpy(ctx, np.array([n], dtype=np.uint32))
    ctx.extract_descriptors(d_img.ptr, w, h, src.shape[1], d_pts.ptr, len(pts), None, d_cnt.ptr, sub, frac_bits)
    got = d_pts.to_numpy(SIFT_POINT_DTYPE, (len(pts),))
    ok = np.isfinite(want["data"][:n]).all(axis=1)
    np.testing.assert_array_equal(np.isfinite(got["data"][:n]).all(axis=1), ok)
    assert ok.mean() > 0.999
    l2 = np.linalg.norm(want["data"][:n][ok].astype(np.float64) - got["data"][:n][ok].astype(np.float64), axis=1)
    # north_star tolerance: 1e-4 L2 per descriptor -- every descriptor (only the summation order differs)
    assert l2.max() < 1e-4, np.sort(l2)[-5:]
    # (the typical distance: a few 1e-6 since the angle coordinate became a degree-4 fit within 4e-6 of a bin -- part of
    # the 1e-4 tolerance spent on purpose, sift_keypoints.hip: desc_angle_bins; it was < 1e-6 with the exact-to-the-ulp form)
    assert np.median(l2) < 1e-5, np.median(l2)
    print("descriptor L2 distance to the oracle: median %.3g, max %.3g" % (np.median(l2), l2.max()))
    # unit norm, and the in-place scaling by subsampling (cuSIFT_D.cu:292-296)
    np.testing.assert_allclose(np.linalg.norm(got["data"][:n][ok], axis=1), 1.0, atol=1e-5)
    np.testing.assert_array_equal(want["coords2D"][:n], got["coords2D"][:n])
    np.testing.assert_array_equal(want["scale"][:n], got["scale"][:n])
    # points beyond the counter are not touched
    np.testing.assert_array_equal(pts["data"][n:], got["data"][n:])


@pytest.mark.parametrize("frac_bits", [0, 8])
def test_hip_descriptors_vs_float64_anchor(ctx, oracle, gray1, frac_bits):
    """The HIP descriptors against the INDEPENDENT float64 evaluation of cuSIFT_D.cu:184-297 (tests/test_float64_anchor.py:
    numpy, written from the reference source, sharing nothing with the oracle or the kernels) -- with the bars the
    oracle itself is held to there.  Since round 4 the kernel no longer follows the oracle's summation order and takes
    sqrt / quotient / rsqrt from single instructions: this test says the result is as close to the algorithm's exact
    value as the oracle's is, not merely close to the oracle."""
    from test_float64_anchor import descriptors_f64

    w, h = 640, 480
    src = pitched(gray1)
    dog = oracle.laplace_multi(src, w, h, 0.0)
    pts, n = oracle.find_points_multi(dog, w, h, 0.5, 10.0, 1.0, 16384)
    assert n > 1500
    oracle.compute_orientations(src, w, h, pts, 0, n, frac_bits)
    before = pts[:n].copy()
    fin = np.isfinite(before["orientation"])
    d_img = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    d_cnt = DeviceBuffer.from_numpy(ctx, np.array([n], dtype=np.uint32))
    ctx.extract_descriptors(d_img.ptr, w, h, src.shape[1], d_pts.ptr, len(pts), None, d_cnt.ptr, 1.0, frac_bits)
    got = d_pts.to_numpy(SIFT_POINT_DTYPE, (len(pts),))["data"][:n].astype(np.float64)
    for b in (d_img, d_pts, d_cnt):
        b.free()
    want, at_pi = descriptors_f64(gray1, before["coords2D"][:, 0].astype(np.float64),
                                  before["coords2D"][:, 1].astype(np.float64), before["scale"].astype(np.float64),
                                  before["orientation"].astype(np.float64), frac_bits, want_flags=True)
    ok = fin & np.isfinite(want).all(axis=1)
    assert ok.mean() > 0.999
    l2 = np.linalg.norm(got - want, axis=1)
    smooth = ok & ~at_pi
    assert smooth.mean() > 0.5
    if frac_bits == 0:
        assert np.median(l2[smooth]) < 3e-5 and l2[smooth].max() < 3e-4, (np.median(l2[smooth]), l2[smooth].max())
    else:
        # (the bars of tests/test_float64_anchor.py::test_descriptor_restatement_vs_float64: with fixed-point weights a
        # flipped fraction is worth 1/256 of a pixel difference, and at most two keypoints may combine it with a 180-degree
        # sample that the float64 flag cannot see -- one exists on the fixture, 0.059 away)
        top = np.sort(l2[smooth])
        assert np.median(top) < 5e-4 and top[-3] < 3e-3 and top[-1] < 0.1, (np.median(top), top[-5:])
    assert l2[ok].max() < 0.2
    # and the oracle is no closer to the float64 value than the kernel is (same keypoints, same bars)
    ref = pts.copy()
    oracle.extract_descriptors(src, w, h, ref, 0, n, 1.0, frac_bits)
    l2_oracle = np.linalg.norm(ref["data"][:n].astype(np.float64) - want, axis=1)
    assert np.median(l2[smooth]) < 1.05 * np.median(l2_oracle[smooth]) + 1e-7


def test_descriptor_quirk_paths(ctx, oracle):
    """Hand-placed keypoints that force the reference's index-overflow paths:
    rows identical + orientation 0 -> dy == +0 exactly, dx < 0 on falling ramps -> atan2f == +pi ->
    angle index 8 spills into the next cell's bin 0; large scales leave the LDS patch (global path);
    keypoints at the image corners exercise the clamped footprints."""
    w, h = 256, 192
    x = np.arange(w, dtype=np.float32)
    row = np.where((x // 16) % 2 == 0, 200 - 8 * (x % 16), 72 + 8 * (x % 16))  # saw-tooth ramps
    img = np.tile(row, (h, 1)).astype(np.float32)
    src = pitched(img)
    pts = np.zeros(64, dtype=SIFT_POINT_DTYPE)
    rng = np.random.default_rng(3)
    n = 48
    pts["coords2D"][:n, 0] = rng.uniform(-2, w + 2, n)
    pts["coords2D"][:n, 1] = rng.uniform(-2, h + 2, n)
    pts["scale"][:n] = np.concatenate([rng.uniform(0.8, 2.2, 32), rng.uniform(2.6, 9.0, 16)])
    pts["orientation"][:n] = np.where(np.arange(n) % 3 == 0, 0.0, rng.uniform(0, 360, n))
    pts["coords2D"][0] = (0.0, 0.0)
    pts["coords2D"][1] = (w - 1.0, h - 1.0)
    want = pts.copy()
    oracle.extract_descriptors(src, w, h, want, 0, n, 1.0, 8)
    d_img = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    d_cnt = DeviceBuffer.from_numpy(ctx, np.array([n], dtype=np.uint32))
    ctx.extract_descriptors(d_img.ptr, w, h, src.shape[1], d_pts.ptr, len(pts), None, d_cnt.ptr, 1.0, 8)
    got = d_pts.to_numpy(SIFT_POINT_DTYPE, (len(pts),))
    fin = np.isfinite(want["data"][:n]).all(axis=1)
    assert fin.sum() >= n - 4
    np.testing.assert_array_equal(np.isfinite(got["data"][:n]).all(axis=1), fin)
    l2 = np.linalg.norm(want["data"][:n][fin].astype(np.float64) - got["data"][:n][fin].astype(np.float64), axis=1)
    assert l2.max() < 1e-4, np.sort(l2)[-5:]


def test_first_offset_restricts_the_range(ctx, oracle, gray1):
    w, h = 640, 480
    src, pts, n = oracle_octave_points(oracle, gray1, w, h, 0.0, 1.0, 1.0)
    first = n // 2
    d_img = DeviceBuffer.from_numpy(ctx, src)
    pts["orientation"] = -7.0
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    d_cnt = DeviceBuffer.from_numpy(ctx, np.array([n], dtype=np.uint32))
    d_fst = DeviceBuffer.from_numpy(ctx, np.array([first], dtype=np.uint32))
    ctx.compute_orientations(d_img.ptr, w, h, src.shape[1], d_pts.ptr, len(pts), d_fst.ptr, d_cnt.ptr, 8)
    got = d_pts.to_numpy(SIFT_POINT_DTYPE, (len(pts),))
    assert (got["orientation"][:first] == -7.0).all()
    assert (got["orientation"][first:n] != -7.0).all()
    assert (got["orientation"][n:] == -7.0).all()


def test_rootsift_matches_oracle(ctx, oracle, gray1):
    pts = oracle.extract(gray1, num_octaves=3, peak_thresh=1.0, max_pts=4096)
    n = len(pts)
    want = pts.copy()
    oracle.rootsift(want, n)
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    ctx.rootsift(d_pts.ptr, n)
    got = d_pts.to_numpy(SIFT_POINT_DTYPE, (n,))
    np.testing.assert_allclose(got["data"], want["data"], rtol=2e-7, atol=1e-9)
    np.testing.assert_allclose((got["data"].astype(np.float64) ** 2).sum(axis=1), 1.0, atol=1e-5)


# ------------------------------------------------------------------------------------------------
# End to end
# ------------------------------------------------------------------------------------------------
def gpu_extract(ctx, img, **kw):
    prm = capi.default_params(**kw)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    d_pts.zero()
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
    return h_pts[:n]


def canonical(p):
    return p[np.lexsort((p["scale"], p["coords2D"][:, 0], p["coords2D"][:, 1], -p["subsampling"]))]


@pytest.mark.parametrize("fused", [1, 0])
def test_root_sift_epilogue_equals_second_pass(ctx, oracle, gray1, fused):
    """cusift_params.root_sift (the fusion the reference leaves as a TODO, cuSIFT.cu:122-134,376-379): RootSIFT as
    the descriptor kernel's epilogue == Extract followed by ConvertSiftToRootSift, bit for bit, on both drivers;
    and it is the oracle's RootSIFT of the plain descriptors."""
    kw = dict(num_octaves=4, peak_thresh=0.5, max_pts=8192, fused_detect=fused)
    plain = canonical(gpu_extract(ctx, gray1, **kw))
    rooted = canonical(gpu_extract(ctx, gray1, root_sift=1, **kw))
    assert len(plain) == len(rooted) > 1000
    for f in ("coords2D", "scale", "orientation", "sharpness", "edgeness", "subsampling"):
        assert np.array_equal(plain[f], rooted[f]), f
    d_pts = DeviceBuffer.from_numpy(ctx, plain)
    ctx.rootsift(d_pts.ptr, len(plain))
    second_pass = d_pts.to_numpy(SIFT_POINT_DTYPE, (len(plain),))
    assert np.array_equal(second_pass["data"], rooted["data"])
    want = plain.copy()
    oracle.rootsift(want, len(want))
    np.testing.assert_allclose(rooted["data"], want["data"], rtol=2e-7, atol=1e-9)
    np.testing.assert_allclose((rooted["data"].astype(np.float64) ** 2).sum(axis=1), 1.0, atol=1e-5)


def compare_sets(want, got):
    """Set-wise comparison of two extractions of the same image: EVERY keypoint, no tolerated fraction.

    north_star: locations / scales / orientations within 1e-3, descriptors within 1e-4 L2.  Demanded here: the same
    point set with bit-identical location, scale, sharpness, edgeness and orientation (the kernels and the oracle
    evaluate the same operations, transcendental functions included: sift_math.h), NaN orientations / descriptors
    (flat patches) in the same places, and every finite descriptor within 1e-4 L2 (summation order only)."""
    assert len(want) == len(got), (len(want), len(got))
    a, b = canonical_order(want), canonical_order(got)
    np.testing.assert_array_equal(a["subsampling"], b["subsampling"])
    np.testing.assert_array_equal(a["coords2D"], b["coords2D"])
    np.testing.assert_array_equal(a["scale"], b["scale"])
    np.testing.assert_array_equal(a["sharpness"], b["sharpness"])
    np.testing.assert_array_equal(a["edgeness"], b["edgeness"])
    np.testing.assert_array_equal(a["orientation"], b["orientation"])  # NaN == NaN positionally
    fin = np.isfinite(a["data"]).all(axis=1)
    np.testing.assert_array_equal(np.isfinite(b["data"]).all(axis=1), fin)
    assert fin.mean() > 0.999
    l2 = np.linalg.norm(a["data"][fin].astype(np.float64) - b["data"][fin].astype(np.float64), axis=1)
    assert l2.max() < 1e-4, (l2.max(), int((l2 >= 1e-4).sum()))
    return l2


def test_extract_fixture_matches_oracle(ctx, oracle, gray1):
    want = oracle.extract(gray1, **REF_PARAMS)
    got = gpu_extract(ctx, gray1, **REF_PARAMS)
    assert len(got) == 9508
    # octave blocks coarsest first, like the reference (cuSIFT.cu:190-196)
    assert np.all(np.diff(got["subsampling"]) <= 0)
    compare_sets(want, got)


@pytest.mark.parametrize("which", ["cusift1_check", "cusift1"])
def test_extract_fixture_vs_reference_golden(ctx, gray1, golden_check, golden_run2, which, record_property):
    """The HIP path itself against BOTH of the reference's golden files, all 4096 rows of each: the gates of the oracle's
    pin (tests/parity_utils.golden_gates -- location + scale, and orientation on the coarse AND the octave-0 rows)."""
    got = gpu_extract(ctx, gray1, **REF_PARAMS)
    res = golden_gates(golden_check if which == "cusift1_check" else golden_run2, got, "HIP path")
    for k, v in res.items():
        record_property(k, v)


def test_canonical_sort_makes_runs_identical_arrays(ctx, gray1):
    """cusift_sort_points_host: the append order inside an octave is racy (as in the reference); after the canonical
    sort two runs give byte-identical arrays, and the order is the one the tests' canonical_order() uses."""
    runs = []
    for _ in range(3):
        got = gpu_extract(ctx, gray1, **REF_PARAMS).copy()
        capi.sort_points(got)
        runs.append(got)
    assert runs[0].tobytes() == runs[1].tobytes() == runs[2].tobytes()
    ref = canonical_order(runs[0])
    for f in ("subsampling", "coords2D", "scale", "orientation"):
        np.testing.assert_array_equal(runs[0][f], ref[f])
    assert np.all(np.diff(runs[0]["subsampling"]) <= 0)


def test_extract_saturates_like_the_reference(ctx, gray1):
    """maxPts=4096 as in test/detector.cpp:41: numPts == maxPts, coarse octaves complete (1555 rows)."""
    prm = dict(REF_PARAMS)
    prm["max_pts"] = 4096
    got = gpu_extract(ctx, gray1, **prm)
    assert len(got) == 4096
    assert int((got["subsampling"] >= 2.0).sum()) == 1555
    assert (got["subsampling"][:1555] >= 2.0).all() and (got["subsampling"][1555:] == 1.0).all()


@pytest.mark.parametrize("cfg", ["C1", "initblur1"])
def test_extract_configs(ctx, oracle, gray1, cfg):
    if cfg == "C1":  # BASELINE configs[0]: 640x480, 3 octaves
        kw = dict(num_octaves=3, init_blur=0.0, peak_thresh=0.1, max_pts=16384)
    else:  # degenerate initBlur (>= first level sigma): documented identity rule
        kw = dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=65536)  # must not saturate max_pts
    want = oracle.extract(gray1, **kw)
    got = gpu_extract(ctx, gray1, **kw)
    assert len(want) > 100
    assert np.isfinite(got["coords2D"]).all()
    compare_sets(want, got)


def test_extract_1080p_matches_oracle(ctx, oracle):
    """BASELINE configs[1]: single 1920x1080, 5 octaves, initBlur 1.0, thresh 3.0 (synthetic tile image)."""
    img = synth.tile(1000, preblur=1.0)
    kw = dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=32768)
    want = oracle.extract(img, **kw)
    got = gpu_extract(ctx, img, **kw)
    assert len(want) > 1000
    compare_sets(want, got)


@pytest.mark.parametrize("case", ["fixture", "1080p"])
def test_extract_vs_glibc_oracle(ctx, gray1, record_property, case):
    """The HIP path against the oracle's GLIBC build (Oracle("libm"): the same restatement on glibc's expf / exp2f /
    atan2f / sinf / cosf instead of the written-out functions of sift_math.h that the kernels share with the default
    oracle build).  Closes the loop "bit-identical to an oracle that compiles the product's own math header" on the GPU:
    same point set, location / sharpness / edgeness equal, scale within a few ulp, orientation within 1e-3 degrees for
    >= 99 % and descriptors within 1e-4 L2 for >= 98 % of the keypoints (the bars test_shared_math_build_agrees_with_
    glibc_build holds between the two oracle builds on the CPU: a last bit of one of five functions occasionally decides
    a 1/256 texture-fraction step or a histogram bin).  The fixture with the parameters of test/detector.cpp:37-49, and
    one 1080p image with the benchmark's (BASELINE configs[1])."""
    from oracle_binding import Oracle

    if case == "fixture":
        img, kw = gray1, dict(REF_PARAMS)
    else:
        img, kw = synth.tile(1000, preblur=1.0), dict(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0,
                                                      max_pts=32768)
    want = canonical_order(Oracle("libm").extract(img, **kw))
    got = canonical_order(gpu_extract(ctx, img, **kw))
    assert len(got) == len(want) and len(got) > 2000
    np.testing.assert_array_equal(got["subsampling"], want["subsampling"])
    np.testing.assert_array_equal(got["coords2D"], want["coords2D"])
    np.testing.assert_array_equal(got["sharpness"], want["sharpness"])
    np.testing.assert_array_equal(got["edgeness"], want["edgeness"])
    np.testing.assert_allclose(got["scale"], want["scale"], rtol=1e-6)
    assert np.array_equal(np.isfinite(got["orientation"]), np.isfinite(want["orientation"]))
    d = ang_diff(got["orientation"].astype(np.float64), want["orientation"].astype(np.float64))
    fin = np.isfinite(d)
    l2 = np.linalg.norm(got["data"][fin].astype(np.float64) - want["data"][fin], axis=1)
    ok = np.isfinite(l2)
    frac_ori, frac_desc = float((d[fin] < 1e-3).mean()), float((l2[ok] < 1e-4).mean())
    record_property("ori_within_1e-3_deg", frac_ori)
    record_property("desc_within_1e-4", frac_desc)
    record_property("desc_median_l2", float(np.median(l2[ok])))
    print("HIP vs glibc oracle (%s): %d keypoints, orientation within 1e-3 deg %.4f, descriptors within 1e-4 L2 %.4f "
          "(median %.2e)" % (case, len(got), frac_ori, frac_desc, float(np.median(l2[ok]))))
    assert fin.mean() > 0.999 and frac_ori >= 0.99
    assert frac_desc >= 0.98 and np.median(l2[ok]) < 1e-5


def test_extract_large_image_matches_oracle(ctx, oracle):
    """A 4096 x 3072 image (12.6 Mpx, 6 octaves): the strip/chunk geometry far from the 1080p case."""
    img = synth.tile(31, 4096, 3072, preblur=0.8)
    kw = dict(num_octaves=6, init_blur=0.8, peak_thresh=3.5, edge_thresh=10.0, max_pts=262144)
    want = oracle.extract(img, **kw)
    got = gpu_extract(ctx, img, **kw)
    assert len(want) > 10000
    compare_sets(want, got)


@pytest.mark.parametrize("w,h,n_oct,blur,thresh", [(1366, 768, 5, 0.0, 2.0), (1000, 750, 5, 1.0, 3.0),
                                                    (1920, 1080, 7, 1.0, 3.0), (1027, 301, 6, 0.5, 2.0)])
def test_extract_ragged_widths_match_oracle(ctx, oracle, w, h, n_oct, blur, thresh):
    """Widths whose octaves are not multiples of 4 (1366 -> 683 -> 341 ...; 1000 -> 250 -> 125; 1080p x 7 octaves ->
    w = 30, 15) run the same fast kernels -- the partial last column group is clamped and replicated in registers --
    and every octave takes the fused detection (no demotion of the batch to the generic path)."""
    img = synth.tile(300 + w, w, h, preblur=blur)
    kw = dict(num_octaves=n_oct, init_blur=blur, peak_thresh=thresh, edge_thresh=10.0, max_pts=65536)
    want = oracle.extract(img, **kw)
    assert len(want) > 500
    prm = capi.default_params(**kw)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    ctx.timing_enable(True)
    ctx.timing_reset()
    n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
    t = ctx.timing_read()
    ctx.timing_enable(False)
    if not os.environ.get("CUSIFT_FORCE_GENERIC"):
        # (with the stage timers on the driver keeps the launch-per-octave sequence)
        assert t["detect_multi"][1] == n_oct and t["laplace_multi"][1] == 0 and t["describe_all"][1] == 1, t
    compare_sets(want, h_pts[:n])
    d_pts.free()


def test_extract_device_image_and_odd_size(ctx, oracle):
    """Legacy ExtractSift path: image already on the device in a caller-pitched buffer; odd w/h."""
    w, h = 333, 251
    img = rand_image(h, w, 11)
    src = pitched(img)
    kw = dict(num_octaves=4, init_blur=0.0, peak_thresh=2.0, max_pts=8192)
    want = oracle.extract(img, **kw)
    prm = capi.default_params(**kw)
    d_img = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n = ctx.extract(d_img.ptr, w, h, src.shape[1], prm, d_pts.ptr, h_pts)
    assert n == len(want) and n > 50
    compare_sets(want, h_pts[:n])


def test_graph_replay_equals_eager(ctx, gray1):
    """cusift_graph_*: the recorded launch sequence replays to the same SiftData as the eager driver, also after the
    caller has put a new frame into the same input buffer (the use it exists for)."""
    h, w = gray1.shape
    kw = dict(num_octaves=5, init_blur=0.0, peak_thresh=0.5, max_pts=8192)
    prm = capi.default_params(**kw)
    frames = [gray1, np.roll(gray1, (31, 77), axis=(0, 1)), gray1[::-1].copy()]
    src = pitched(frames[0])
    p = src.shape[1]
    d_img = DeviceBuffer.from_numpy(ctx, src)
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    d_cnt = DeviceBuffer(ctx, 4)

    def read():
        ctx.synchronize()
        n = min(int(d_cnt.to_numpy(np.uint32, (1,))[0]), prm.max_pts)
        return canonical(d_pts.to_numpy(SIFT_POINT_DTYPE, (prm.max_pts,))[:n])

    graph = ctx.record_graph(d_img.ptr, 1, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
    assert graph.nodes >= 3  # the ScaleDown chain, the detection of all octaves, description: one launch each
    for f in frames:
        ctx.h2d(d_img.ptr, pitched(f))
        d_pts.zero()
        ctx.extract_batch(d_img.ptr, 1, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
        eager = read()
        assert len(eager) > 500
        for _ in range(2):
            d_pts.zero()
            graph.launch()
            again = read()
            assert len(again) == len(eager)
            for fld in ("coords2D", "scale", "orientation", "sharpness", "edgeness", "subsampling", "data"):
                assert np.array_equal(again[fld], eager[fld]), fld
    graph.close()
    # a recording is tied to the arena it was made with: growing the arena invalidates it, loudly
    with capi.Context(0) as fresh:
        d_img2 = DeviceBuffer.from_numpy(fresh, src)
        d_pts2 = DeviceBuffer(fresh, prm.max_pts * 588)
        d_cnt2 = DeviceBuffer(fresh, 4)
        g2 = fresh.record_graph(d_img2.ptr, 1, w, h, p, h * p, prm, d_pts2.ptr, d_cnt2.ptr)
        g2.launch()
        fresh.synchronize()
        fresh.reserve(4, 2048, 2048, prm)
        with pytest.raises(capi.CusiftError):
            g2.launch()
        g2.close()
        for b in (d_img2, d_pts2, d_cnt2):
            b.free()
    # the null stream cannot be captured: refused, not crashed
    with capi.Context(0, stream=0) as null_ctx:
        with pytest.raises(capi.CusiftError):
            null_ctx.record_graph(d_img.ptr, 1, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)


def test_extract_batch_equals_single(ctx, oracle, gray1):
    """Batch form: n images in one launch sequence == n single extractions (set-wise)."""
    imgs = [gray1, np.roll(gray1, (13, 57), axis=(0, 1)), gray1[::-1, ::-1].copy()]
    n = len(imgs)
    h, w = gray1.shape
    kw = dict(num_octaves=4, init_blur=0.0, peak_thresh=0.5, max_pts=8192)
    prm = capi.default_params(**kw)
    stack = np.stack([pitched(i) for i in imgs])
    p = stack.shape[2]
    d_imgs = DeviceBuffer.from_numpy(ctx, stack)
    d_pts = DeviceBuffer(ctx, n * prm.max_pts * 588)
    d_cnt = DeviceBuffer(ctx, 4 * n)
    ctx.extract_batch(d_imgs.ptr, n, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
    ctx.synchronize()
    cnt = d_cnt.to_numpy(np.uint32, (n,))
    pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (n, prm.max_pts))
    for i in range(n):
        want = oracle.extract(imgs[i], **kw)
        assert int(cnt[i]) == len(want)
        compare_sets(want, pts[i, : cnt[i]])


def test_full_size_batch_properties(ctx):
    """BASELINE configs[2] shape at reduced count (8 x 1080p): size-independent properties --
    identical images give identical point sets; a constant image gives none; counts are stable
    across two runs (the pipeline has no cross-image state)."""
    n, w, h = 8, 1920, 1080
    base = synth.tile(1000, preblur=1.0)
    other = synth.tile(1001, preblur=1.0)
    imgs = np.stack([pitched(base) if i % 2 == 0 else pitched(other) for i in range(n)])
    imgs[7][:] = 77.0
    p = imgs.shape[2]
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, max_pts=32768)
    d_imgs = DeviceBuffer.from_numpy(ctx, imgs)
    d_pts = DeviceBuffer(ctx, n * prm.max_pts * 588)
    d_cnt = DeviceBuffer(ctx, 4 * n)
    runs = []
    for _ in range(2):
        ctx.extract_batch(d_imgs.ptr, n, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
        ctx.synchronize()
        runs.append(d_cnt.to_numpy(np.uint32, (n,)).copy())
    np.testing.assert_array_equal(runs[0], runs[1])
    cnt = runs[0]
    assert cnt[7] == 0
    assert cnt[0] == cnt[2] == cnt[4] == cnt[6] and cnt[1] == cnt[3] == cnt[5] and cnt[0] > 1000
    pts = d_pts.to_numpy(SIFT_POINT_DTYPE, (n, prm.max_pts))
    a, b = canonical_order(pts[0, : cnt[0]]), canonical_order(pts[2, : cnt[2]])
    np.testing.assert_array_equal(a["coords2D"], b["coords2D"])
    np.testing.assert_array_equal(a["orientation"], b["orientation"])
    np.testing.assert_array_equal(a["data"], b["data"])  # one wave per keypoint: reproducible sums


def test_pack_points_matches_torch_expression(ctx, gray1):
    """cusift_pack_points == the boolean-mask packing used on CPU (cusift_amd.dist.pack_points)."""
    import torch

    from cusift_amd.batch import BatchExtractor
    from cusift_amd.dist import pack_points

    imgs = np.stack([gray1, np.roll(gray1, (9, 31), axis=(0, 1)), np.full_like(gray1, 50.0), gray1[::-1].copy()])
    ex = BatchExtractor(4, 640, 480, num_octaves=3, peak_thresh=1.0, max_pts=2048)
    pts, cnt = ex.extract(ex.images_from_numpy(imgs))
    torch.cuda.synchronize()
    assert int(cnt[2]) == 0 and int(cnt[0]) > 100
    want_p, want_v = pack_points(pts, cnt, ex.max_pts)
    for stream in (None, torch.cuda.Stream()):
        got_p, got_v = ex.make_packer(stream)(pts, cnt, ex.max_pts)
        torch.cuda.synchronize()
        assert torch.equal(got_v.cpu(), want_v.cpu())
        assert torch.equal(got_p.cpu(), want_p.cpu())
    # saturated counters are clamped to max_pts
    cnt2 = cnt.clone()
    cnt2[1] = 10 ** 6
    got_p, got_v = ex.make_packer()(pts, cnt2, ex.max_pts)
    assert int(got_v[1]) == ex.max_pts and got_p.shape[0] == int(got_v.sum())
    ex.close()


def test_stage_timers_report_every_stage(ctx, gray1):
    import os
    if os.environ.get("CUSIFT_FORCE_GENERIC"):
        pytest.skip("CUSIFT_FORCE_GENERIC disables the fused launches this test counts")
    d_pts = DeviceBuffer(ctx, 4096 * 588)
    for fused in (1, 0):
        prm = capi.default_params(num_octaves=3, peak_thresh=1.0, max_pts=4096, fused_detect=fused)
        ctx.timing_enable(True)
        ctx.timing_reset()
        ctx.extract_host(gray1, prm, d_pts.ptr, None)
        t = ctx.timing_read()
        ctx.timing_enable(False)
        assert t["scale_down"][1] == 2
        if fused:  # 3 fused detections (a launch each while the stage timers are on), then ONE orientation+descriptor
            #        launch over all octaves
            assert t["detect_multi"][1] == 3 and t["describe_all"][1] == 1
            assert t["laplace_multi"][1] == 0 and t["find_points_multi"][1] == 0 and t["extract_descriptors"][1] == 0
        else:      # the reference's per-octave stage sequence
            assert t["detect_multi"][1] == 0 and t["describe_all"][1] == 0
            assert t["laplace_multi"][1] == 3 and t["find_points_multi"][1] == 3
            assert t["compute_orientations"][1] == 3 and t["extract_descriptors"][1] == 3
        assert t["total"][1] == 1 and t["total"][0] > 0
        assert all(ms >= 0 for ms, _ in t.values())


def test_errors_are_reported_not_fatal(ctx):
    prm = capi.default_params()
    with pytest.raises(capi.CusiftError, match="missing data"):
        ctx.extract_batch(None, 1, 64, 64, 128, 64 * 128, prm, None, None)
    with pytest.raises(capi.CusiftError):
        ctx.scale_down(None, 128, None, 64, 64, 128)
    bad = capi.default_params(max_pts=0)
    d = DeviceBuffer(ctx, 1024)
    with pytest.raises(capi.CusiftError, match="max_pts"):
        ctx.extract_batch(d.ptr, 1, 8, 8, 8, 64, bad, d.ptr, d.ptr)
    assert C.c_char_p(capi.lib().cusift_last_error()).value


def test_pipelined_extractor_equals_single_stream(ctx, gray1):
    """PipelinedExtractor (consecutive batches alternating over two streams): every batch's SiftData equals what one
    BatchExtractor gives for that batch, bit for bit (canonical order)."""
    import torch

    from cusift_amd.batch import BatchExtractor, PipelinedExtractor

    kw = dict(num_octaves=4, peak_thresh=1.0, max_pts=4096)
    batches = [np.stack([np.roll(gray1, (7 * b + 3 * i, 11 * b + 5 * i), axis=(0, 1)) for i in range(3)])
               for b in range(5)]
    single = BatchExtractor(3, 640, 480, **kw)
    want = []
    for imgs in batches:
        single.extract(single.images_from_numpy(imgs))
        want.append([canonical(p) for p in single.to_host()])
    single.close()

    pipe = PipelinedExtractor(3, 640, 480, n_streams=2, n_slots=1, **kw)
    dev = [pipe.images_from_numpy(imgs) for imgs in batches]
    torch.cuda.synchronize()
    for b, d_imgs in enumerate(dev):
        pts, cnt, done = pipe.submit(d_imgs)
        done.synchronize()  # outputs of a stream are reused two submits later: read them back first
        counts = torch.clamp(cnt, max=pipe.max_pts).cpu().numpy()
        for i in range(3):
            got = canonical(pts[i, : int(counts[i])].cpu().numpy().view(SIFT_POINT_DTYPE).reshape(-1))
            assert len(got) == len(want[b][i]) > 100
            for f in ("coords2D", "scale", "orientation", "data"):
                assert np.array_equal(got[f], want[b][i][f]), (b, i, f)
    # and without a wait in between (both streams busy): the last two batches are still intact
    outs = [pipe.submit(d) for d in dev[-2:]]
    pipe.synchronize()
    for (pts, cnt, _), wb in zip(outs, want[-2:]):
        counts = torch.clamp(cnt, max=pipe.max_pts).cpu().numpy()
        for i in range(3):
            got = canonical(pts[i, : int(counts[i])].cpu().numpy().view(SIFT_POINT_DTYPE).reshape(-1))
            assert np.array_equal(got["data"], wb[i]["data"])
    pipe.close()


def test_torch_still_sees_the_gpu_when_imported_after_the_library():
    """Load order: libcusift_amd.so first, torch second.  The process can hold one libamdhip64.so.7; capi.lib() loads
    the copy a PyTorch-ROCm wheel ships (when there is one) so that a later `import torch` finds its own runtime."""
    import subprocess
    import sys

    # allocation + copy only: no torch kernel is launched, so the child does not have to load torch's code objects
    # (minutes on a box whose page cache is cold)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cusift_amd import capi\n"
            "c = capi.Context(0); b = capi.DeviceBuffer(c, 1 << 20)\n"
            "import torch\n"
            "assert torch.cuda.is_available() and torch.cuda.device_count() >= 1\n"
            "t = torch.empty(8, device='cuda'); t.copy_(torch.full((8,), 1.0)); torch.cuda.synchronize()\n"
            "print(float(t.cpu().sum()))\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("8.0"), out.stdout[-500:] + out.stderr[-1500:]
