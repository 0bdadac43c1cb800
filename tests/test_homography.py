"""RANSAC homography: FindHomography, extras/homography.cu:182-269.

PARITY UNPINNED: the reference holds no test or fixture for it (main.cpp calls it, nothing checks it), so the
oracle restatement is anchored on what the algorithm must deliver (a planted homography is recovered, its inliers
counted) and the HIP path is compared with the oracle bit for bit on the same samples."""
import numpy as np
import pytest

from oracle_binding import SIFT_POINT_DTYPE


def planted(n_in=400, n_out=250, seed=3, noise=0.3):
    """Matched keypoints of a 1280x960 image pair related by a known homography, plus gross outliers."""
    rng = np.random.default_rng(seed)
    H = np.array([[0.92, -0.11, 37.0], [0.08, 1.05, -21.0], [2.1e-5, -3.4e-5, 1.0]])
    n = n_in + n_out
    pts = np.zeros(n, dtype=SIFT_POINT_DTYPE)
    xy = rng.uniform([0, 0], [1280, 960], size=(n, 2))
    proj = np.c_[xy, np.ones(n)] @ H.T
    proj = proj[:, :2] / proj[:, 2:]
    proj[:n_in] += rng.normal(0, noise, size=(n_in, 2))
    proj[n_in:] = rng.uniform([0, 0], [1280, 960], size=(n_out, 2))
    perm = rng.permutation(n)
    pts["coords2D"] = xy[perm].astype(np.float32)
    pts["match_xpos"] = proj[perm, 0].astype(np.float32)
    pts["match_ypos"] = proj[perm, 1].astype(np.float32)
    pts["score"] = 0.9
    pts["ambiguity"] = 0.5
    inlier = np.zeros(n, dtype=bool)
    inlier[:n_in] = True
    return pts, H, inlier[perm]


def draw_samples(n_valid, loops, seed):
    """4 distinct indices per hypothesis (the reference draws with rand() and re-draws duplicates, :222-235)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((4, loops), dtype=np.int32)
    for l in range(loops):
        out[:, l] = rng.choice(n_valid, size=4, replace=False)
    return out


def test_oracle_recovers_a_planted_homography(oracle):
    pts, H, inlier = planted()
    rand_pts = draw_samples(len(pts), 1008, 7)
    hom, n_match, best, all_h, all_c = oracle.find_homography(pts, rand_pts, thresh=3.0)
    assert hom[8] == 1.0 and best == int(np.argmax(all_c)) and n_match == all_c.max()
    # RANSAC from 4-point samples: nearly all planted inliers are within 3 px of the winning hypothesis ...
    assert inlier.sum() * 0.9 <= n_match <= inlier.sum() + 12
    # ... and the winner maps the image corners like the planted homography does (within a few pixels)
    corners = np.array([[0, 0, 1], [1280, 0, 1], [0, 960, 1], [1280, 960, 1]], dtype=np.float64)
    a = corners @ hom.reshape(3, 3).astype(np.float64).T
    b = corners @ H.T
    assert np.abs(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:]).max() < 4.0
    # inlier counts are what a float64 evaluation of the same test gives, up to borderline points
    x1, y1 = pts["coords2D"][:, 0].astype(np.float64), pts["coords2D"][:, 1].astype(np.float64)
    for l in (best, 0, 17):
        h8 = all_h[:, l].astype(np.float64)
        den = h8[6] * x1 + h8[7] * y1 + 1.0
        ex = pts["match_xpos"] * den - (h8[0] * x1 + h8[1] * y1 + h8[2])
        ey = pts["match_ypos"] * den - (h8[3] * x1 + h8[4] * y1 + h8[5])
        ref = int((ex * ex + ey * ey < 9.0 * den * den).sum())
        assert abs(ref - int(all_c[l])) <= 2, (l, ref, all_c[l])


def test_oracle_four_exact_correspondences_are_interpolated(oracle):
    """The 8x8 solve itself: a hypothesis reproduces its own 4 samples -- as far as a float32 LU of the
    unnormalised DLT system (condition number ~1e9 at these coordinates) can: median error ~1e-3 px, worst < 1 px."""
    pts, _, _ = planted(n_in=40, n_out=0, noise=0.0)
    rand_pts = draw_samples(len(pts), 64, 1)
    _, _, _, all_h, all_c = oracle.find_homography(pts, rand_pts, thresh=1.0)
    x = pts["coords2D"].astype(np.float64)
    errs = []
    for l in range(64):
        h8 = all_h[:, l].astype(np.float64)
        for i in rand_pts[:, l]:
            den = h8[6] * x[i, 0] + h8[7] * x[i, 1] + 1.0
            px = (h8[0] * x[i, 0] + h8[1] * x[i, 1] + h8[2]) / den
            py = (h8[3] * x[i, 0] + h8[4] * x[i, 1] + h8[5]) / den
            errs.append(max(abs(px - pts["match_xpos"][i]), abs(py - pts["match_ypos"][i])))
    assert np.median(errs) < 0.01 and max(errs) < 1.0, (np.median(errs), max(errs))
    assert (all_c >= 4).all()  # every hypothesis counts at least its own samples


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,n_out,loops", [(400, 250, 1008), (9, 0, 16), (3000, 5000, 2000)])
def test_hip_homography_equals_oracle(ctx, oracle, n_in, n_out, loops):
    from cusift_amd.capi import DeviceBuffer

    pts, _, _ = planted(n_in, n_out, seed=n_in)
    rand_pts = draw_samples(len(pts), loops, 11)
    want_h, want_n, want_best, want_all_h, want_all_c = oracle.find_homography(pts, rand_pts, thresh=5.0)
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    hom, n_match, all_h, all_c = ctx.find_homography(d_pts.ptr, len(pts), rand_pts, thresh=5.0, want_all=True)
    assert np.array_equal(all_c, want_all_c)                      # integer work: identical
    assert np.array_equal(all_h.view(np.uint32), want_all_h.view(np.uint32))  # same arithmetic: same bits
    assert n_match == want_n and np.array_equal(hom.view(np.uint32), want_h.view(np.uint32))
    assert n_match == all_c.max() and np.array_equal(hom[:8], all_h[:, int(np.argmax(all_c))])


@pytest.mark.gpu
def test_hip_homography_degenerate_samples_and_bad_indices(ctx, oracle):
    """Collinear / repeated samples give a singular system: the reference's 1e-16 pivot rule produces garbage
    hypotheses, not a crash, and they are the oracle's garbage bit for bit (NaN patterns included)."""
    from cusift_amd import capi
    from cusift_amd.capi import DeviceBuffer

    pts, _, _ = planted(60, 20, seed=5)
    pts["coords2D"][:8, 1] = 100.0  # eight collinear points
    rand_pts = draw_samples(len(pts), 64, 2)
    rand_pts[:, 0] = [0, 1, 2, 3]   # collinear
    rand_pts[:, 1] = [5, 5, 5, 5]   # the same point four times
    want = oracle.find_homography(pts, rand_pts, thresh=5.0)
    d_pts = DeviceBuffer.from_numpy(ctx, pts)
    hom, n_match, all_h, all_c = ctx.find_homography(d_pts.ptr, len(pts), rand_pts, thresh=5.0, want_all=True)
    assert np.array_equal(all_c, want[4]) and np.array_equal(all_h.view(np.uint32), want[3].view(np.uint32))
    bad = rand_pts.copy()
    bad[2, 7] = len(pts)
    with pytest.raises(capi.CusiftError):
        ctx.find_homography(d_pts.ptr, len(pts), bad)
