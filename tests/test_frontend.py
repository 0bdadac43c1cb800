"""Caller-side front-end: 8-bit upload + conversion and the optional 3x3 Gaussian
pre-blur the reference's programs run on the host with OpenCV (main.cpp:300-318, test/detector.cpp:19-27).

The conversion is exact.  The blur's oracle is PARITY UNPINNED (OpenCV is not in this image): the CPU test checks the
restatement against an independent float64 evaluation of the published formula, the GPU test checks the kernel
bit-exactly against the restatement.
"""
import numpy as np
import pytest

from cusift_amd import capi
from cusift_amd.capi import DeviceBuffer


def u8_image(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


def test_oracle_gaussian3x3_follows_the_published_formula(oracle):
    img = u8_image(37, 53, 1).astype(np.float32)
    got = oracle.gaussian3x3(img, 0.5)
    e1 = np.exp(-1.0 / (2 * 0.5 * 0.5))
    k = np.array([e1, 1.0, e1]) / (1 + 2 * e1)
    pad = np.pad(img.astype(np.float64), 1, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    rows = k[0] * pad[:, :-2] + k[1] * pad[:, 1:-1] + k[2] * pad[:, 2:]
    want = k[0] * rows[:-2] + k[1] * rows[1:-1] + k[2] * rows[2:]
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
    # a constant image stays constant to rounding; a single-row / single-column image is handled
    np.testing.assert_allclose(oracle.gaussian3x3(np.full((5, 7), 100.0, np.float32), 0.5), 100.0, atol=1e-4)
    assert oracle.gaussian3x3(img[:1], 0.5).shape == (1, 53)
    assert oracle.gaussian3x3(img[:, :1], 0.5).shape == (37, 1)


def test_oracle_u8_conversion_is_exact(oracle):
    img = u8_image(9, 31, 2)
    np.testing.assert_array_equal(oracle.u8_to_f32(img), img.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(640, 480), (101, 77), (1, 1), (3, 5), (1920, 1080), (1027, 9)])
def test_u8_upload_is_exact(ctx, w, h):
    img = u8_image(h, w, w + h)
    pitch = -(-w // 128) * 128
    d = DeviceBuffer(ctx, h * pitch * 4)
    d.zero()
    ctx.image_u8_h2d(d.ptr, pitch, img)
    got = d.to_numpy(np.float32, (h, pitch))
    np.testing.assert_array_equal(got[:, :w], img.astype(np.float32))
    assert not got[:, w:].any()
    # unaligned pitch takes the scalar path
    d2 = DeviceBuffer(ctx, h * (w + 1) * 4)
    d2.zero()
    ctx.image_u8_h2d(d2.ptr, w + 1, img)
    np.testing.assert_array_equal(d2.to_numpy(np.float32, (h, w + 1))[:, :w], img.astype(np.float32))


@pytest.mark.gpu
def test_u8_batch_conversion_on_device(ctx):
    n, w, h = 3, 200, 45
    imgs = np.stack([u8_image(h, w, 10 + i) for i in range(n)])
    d_src = DeviceBuffer.from_numpy(ctx, imgs)
    pitch = 256
    d_dst = DeviceBuffer(ctx, n * h * pitch * 4)
    d_dst.zero()
    ctx.u8_to_f32(d_dst.ptr, pitch, d_src.ptr, w, h, w, n_images=n)
    ctx.synchronize()
    got = d_dst.to_numpy(np.float32, (n, h, pitch))
    np.testing.assert_array_equal(got[:, :, :w], imgs.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,sigma", [(640, 480, 0.5), (101, 77, 0.5), (1, 9, 0.5), (9, 1, 0.5), (2, 2, 0.8),
                                       (300, 200, 1.0)])
def test_gaussian3x3_bit_exact(ctx, oracle, w, h, sigma):
    img = u8_image(h, w, 3 * w + h).astype(np.float32)
    want = oracle.gaussian3x3(img, sigma)
    pitch = -(-w // 128) * 128
    src = np.zeros((h, pitch), np.float32)
    src[:, :w] = img
    d_src = DeviceBuffer.from_numpy(ctx, src)
    d_dst = DeviceBuffer(ctx, h * pitch * 4)
    d_dst.zero()
    ctx.gaussian3x3(d_dst.ptr, pitch, d_src.ptr, w, h, pitch, sigma)
    ctx.synchronize()
    got = d_dst.to_numpy(np.float32, (h, pitch))
    np.testing.assert_array_equal(got[:, :w], want)
    assert not got[:, w:].any()


@pytest.mark.gpu
def test_frontend_rejects_bad_arguments(ctx):
    d = DeviceBuffer(ctx, 4096)
    with pytest.raises(capi.CusiftError):
        ctx.gaussian3x3(d.ptr, 16, d.ptr, 16, 16, 16, 0.5)  # in place
    with pytest.raises(capi.CusiftError):
        ctx.gaussian3x3(d.ptr, 16, d.ptr + 2048, 16, 16, 16, 0.0)  # sigma <= 0
    with pytest.raises(capi.CusiftError):
        ctx.u8_to_f32(d.ptr, 8, d.ptr + 2048, 16, 4, 16)  # pitch < width


@pytest.mark.gpu
def test_extract_from_u8_equals_extract_from_float(ctx, gray1):
    """The detector test's flow (test/detector.cpp:19-49) with the conversion on the device."""
    img8 = gray1.astype(np.uint8)
    h, w = img8.shape
    prm = capi.default_params(num_octaves=5, init_blur=0.0, peak_thresh=1.0, max_pts=16384)
    pitch = -(-w // 128) * 128
    d_img = DeviceBuffer(ctx, h * pitch * 4)
    d_img.zero()
    ctx.image_u8_h2d(d_img.ptr, pitch, img8)
    d_pts = DeviceBuffer(ctx, prm.max_pts * capi.SIFT_POINT_BYTES)
    h_a = np.zeros(prm.max_pts, dtype=capi.SIFT_POINT_DTYPE)
    h_b = np.zeros(prm.max_pts, dtype=capi.SIFT_POINT_DTYPE)
    na = ctx.extract(d_img.ptr, w, h, pitch, prm, d_pts.ptr, h_a)
    nb = ctx.extract_host(gray1, prm, d_pts.ptr, h_b)
    assert na == nb and na > 500
    from parity_utils import canonical_order
    a, b = canonical_order(h_a[:na]), canonical_order(h_b[:nb])
    for f in ("coords2D", "scale", "orientation", "data"):
        np.testing.assert_array_equal(a[f], b[f])


def test_large_tile_images_equal_the_plain_generator():
    """synth.tile assembles large pre-blurred images (BASELINE configs[4]) from one period of the mirror-tiled pattern;
    the result must be the plain generator's bit for bit -- even and odd tile counts, several shifts."""
    from cusift_amd import synth

    for seed, w, h in ((7, 5300, 3900), (9, 5120, 3840), (11, 5800, 3841)):
        assert np.array_equal(synth.tile(seed, w, h, preblur=1.0), synth._tile_plain(seed, w, h, 1.0)), (seed, w, h)
