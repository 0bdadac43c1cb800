"""Strip tiling (BASELINE configs[4] at reduced size): P virtual ranks on one GPU, halo exchange as device copies.
The union of the per-rank SiftData must equal the whole-image extraction bit for bit."""
import numpy as np
import pytest
import torch

from cusift_amd import capi, synth
from cusift_amd.capi import SIFT_POINT_DTYPE, DeviceBuffer
from cusift_amd.tiling import StripExtractor, StripPlan, run_virtual
from oracle_binding import pitched
from parity_utils import canonical_order

pytestmark = pytest.mark.gpu


def whole_image(ctx, img, prm):
    d_pts = DeviceBuffer(ctx, prm.max_pts * 588)
    h_pts = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
    n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
    return h_pts[:n]


@pytest.mark.parametrize("W,H,P,n_oct,blur,thresh,collapse", [
    (1024, 2048, 4, 4, 1.0, 3.0, 4),   # every octave tiled (512 .. 64 owned rows)
    (512, 1536, 2, 5, 0.0, 2.0, 5),
    (256, 768, 1, 3, 0.0, 2.0, 3),     # one rank: the plain whole image through the band entry points
    (1024, 2048, 4, 7, 1.0, 3.0, 4),   # octaves 4..6 (128, 64, 32 rows) collapse onto rank 0
    (1000, 1531, 3, 6, 0.5, 2.0, 4),   # uneven strips (510/510/511 rows), ragged widths (1000 -> 500 -> 250 -> 125 ...)
    (640, 300, 8, 4, 0.0, 1.0, 0),     # strips thinner than the halo from the start: everything runs on rank 0
])
def test_strips_equal_whole_image(ctx, W, H, P, n_oct, blur, thresh, collapse):
    img = synth.tile(77, W, H, preblur=blur)
    prm = capi.default_params(num_octaves=n_oct, init_blur=blur, peak_thresh=thresh, max_pts=65536)
    want = canonical_order(whole_image(ctx, img, prm))
    assert len(want) > 300
    dev = torch.device("cuda", 0)
    full = torch.from_numpy(img).to(dev)
    exts = [StripExtractor(k, P, W, H, prm, device=dev) for k in range(P)]
    plan = exts[0].plan
    assert plan.collapse == collapse, plan.collapse
    bounds = plan.bounds
    parts = run_virtual(exts, [full[bounds[k]:bounds[k + 1]] for k in range(P)])
    # ownership: every rank only reports keypoints whose detection row it owns (octave rows -> base rows); the
    # collapsed octaves all belong to rank 0
    for k, pts in enumerate(parts):
        tiled = pts[pts["subsampling"] < prm.subsampling * 2 ** plan.collapse]
        if len(tiled):
            yb, slack = tiled["coords2D"][:, 1], 0.51 * tiled["subsampling"].max()
            assert yb.min() >= bounds[k] - slack - tiled["subsampling"].max() and yb.max() < bounds[k + 1] + slack
        if k != plan.root:
            assert len(tiled) == len(pts)
        # coarsest octave first inside each rank's list
        assert np.all(np.diff(pts["subsampling"]) <= 0)
    got = canonical_order(np.concatenate(parts))
    assert len(got) == len(want)
    np.testing.assert_array_equal(got["subsampling"], want["subsampling"])
    np.testing.assert_array_equal(got["coords2D"], want["coords2D"])
    np.testing.assert_array_equal(got["scale"], want["scale"])
    np.testing.assert_array_equal(got["sharpness"], want["sharpness"])
    np.testing.assert_array_equal(got["orientation"], want["orientation"])
    np.testing.assert_array_equal(got["data"], want["data"])
    for e in exts:
        e.close()


def test_footprint_beyond_the_halo_is_an_error_not_a_clamped_read(ctx):
    """A keypoint whose descriptor grid reaches past the halo would sample clamped rows instead of the neighbour's:
    the band kernel counts it and the extractor raises (strict) or reports (strict=False) -- never silently different."""
    W, H, P = 512, 1024, 2
    img = synth.tile(5, W, H)
    prm = capi.default_params(num_octaves=2, init_blur=0.0, peak_thresh=1.0, max_pts=65536)
    dev = torch.device("cuda", 0)
    full = torch.from_numpy(img).to(dev)
    for strict in (True, False):
        exts = [StripExtractor(k, P, W, H, prm, device=dev, halo=8, strict=strict) for k in range(P)]
        b = exts[0].plan.bounds
        if strict:
            with pytest.raises(capi.CusiftError, match="halo"):
                run_virtual(exts, [full[b[k]:b[k + 1]] for k in range(P)])
        else:
            parts = run_virtual(exts, [full[b[k]:b[k + 1]] for k in range(P)])
            assert sum(e.check() for e in exts) > 0 and sum(len(p) for p in parts) > 300
        for e in exts:
            e.close()
    # with the default halo nothing on this image is flagged
    exts = [StripExtractor(k, P, W, H, prm, device=dev) for k in range(P)]
    run_virtual(exts, [full[b[k]:b[k + 1]] for k in range(P)])
    assert all(e.check() == 0 for e in exts)
    for e in exts:
        e.close()


def test_band_entry_points_validate_rows(ctx):
    d = DeviceBuffer(ctx, 1 << 20)
    with pytest.raises(capi.CusiftError, match="row geometry"):
        ctx.detect_band(d.ptr, 128, 64, 128, 10, 60, 10, 74, 0.0, 1.0, 10.0, 1.0, d.ptr, 16, d.ptr)  # band leaves the image
    with pytest.raises(capi.CusiftError, match="halo rows"):
        ctx.detect_band(d.ptr, 128, 40, 128, 10, 100, 12, 48, 0.0, 1.0, 10.0, 1.0, d.ptr, 16, d.ptr)  # 2 rows of halo
    with pytest.raises(capi.CusiftError, match="row geometry"):
        ctx.scale_down_band(d.ptr, 128, 0, 0, 40, d.ptr, 128, 64, 128, 0, 64, 0.5)  # r_end > h/2
