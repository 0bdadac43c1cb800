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


@pytest.mark.parametrize("W,H,P,n_oct,blur,thresh", [(1024, 2048, 4, 4, 1.0, 3.0), (512, 1536, 2, 5, 0.0, 2.0),
                                                      (256, 768, 1, 3, 0.0, 2.0)])
def test_strips_equal_whole_image(ctx, W, H, P, n_oct, blur, thresh):
    img = synth.tile(77, W, H, preblur=blur)
    prm = capi.default_params(num_octaves=n_oct, init_blur=blur, peak_thresh=thresh, max_pts=65536)
    want = canonical_order(whole_image(ctx, img, prm))
    assert len(want) > 300
    dev = torch.device("cuda", 0)
    full = torch.from_numpy(img).to(dev)
    rows = H // P
    exts = [StripExtractor(k, P, W, H, prm, device=dev) for k in range(P)]
    parts = run_virtual(exts, [full[k * rows:(k + 1) * rows] for k in range(P)])
    # ownership: every rank only reports keypoints whose detection row it owns (octave rows -> base rows)
    for k, pts in enumerate(parts):
        if len(pts):
            yb = pts["coords2D"][:, 1]
            assert yb.min() >= k * rows - 0.51 * pts["subsampling"].max() and yb.max() < (k + 1) * rows + 0.51 * pts["subsampling"].max()
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


def test_plan_rejects_bad_geometry():
    with pytest.raises(ValueError):
        StripPlan(1000, 2048, 4, 4)   # width not a multiple of 32
    with pytest.raises(ValueError):
        StripPlan(1024, 2000, 4, 4)   # height not a multiple of 32
    with pytest.raises(ValueError):
        StripPlan(1024, 1024, 8, 5)   # coarsest octave owns 8 rows < halo
    pl = StripPlan(8192, 8192, 8, 5)
    assert pl.own(3, 0) == (3072, 4096) and pl.own(3, 4) == (192, 256)
    assert pl.band(0, 0) == (0, 1024 + 48) and pl.band(7, 4) == (448 - 48, 512)


def test_band_entry_points_validate_rows(ctx):
    d = DeviceBuffer(ctx, 1 << 20)
    with pytest.raises(capi.CusiftError, match="row geometry"):
        ctx.detect_band(d.ptr, 128, 64, 128, 10, 60, 10, 74, 0.0, 1.0, 10.0, 1.0, d.ptr, 16, d.ptr)  # band leaves the image
    with pytest.raises(capi.CusiftError, match="row geometry"):
        ctx.scale_down_band(d.ptr, 128, 0, 0, 40, d.ptr, 128, 64, 128, 0, 64, 0.5)  # r_end > h/2
