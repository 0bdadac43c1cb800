"""Float64 anchors for the parts of the path the reference holds no vector for (SURVEY.md section 8c "NOT pinned").

The reference's golden file carries x, y, scale, orientation only: descriptors, sharpness, edgeness and any non-zero
initBlur have no fixture.  The CPU oracle is the arbiter for them, so it is itself checked here against an INDEPENDENT
evaluation: numpy float64 code written from the reference source (cited per function), sharing nothing with
oracle/sift_oracle.c -- different language, different precision, vectorised over keypoints instead of walking them.
Agreement within float32 rounding says the C restatement computes what the reference's source says.

CPU only (`-m "not gpu"`).
"""
import numpy as np
import pytest

from oracle_binding import pitched
from parity_utils import match_nearest


# ------------------------------------------------------------------------------------------------
# texture model in float64: tex2D<float>, cudaFilterModeLinear, clamp, unnormalised coordinates
# (CUDA programming guide, linear filtering: xB = x - 0.5, i = floor(xB), alpha = frac(xB) [8 fractional bits])
# ------------------------------------------------------------------------------------------------
def tex2d_f64(img, x, y, frac_bits):
    h, w = img.shape
    xb, yb = x - 0.5, y - 0.5
    i, j = np.floor(xb), np.floor(yb)
    a, b = xb - i, yb - j
    i, j = i.astype(np.int64), j.astype(np.int64)
    i0, i1 = np.clip(i, 0, w - 1), np.clip(i + 1, 0, w - 1)
    j0, j1 = np.clip(j, 0, h - 1), np.clip(j + 1, 0, h - 1)
    if frac_bits:
        # the texture unit's weights are fixed-point numbers of the fractions' precision: the product is rounded (half
        # up), the other three follow by subtraction (oracle_tex2d; pinned on the reference's golden orientations)
        q = float(1 << frac_bits)
        A, B = np.floor(a * q + 0.5), np.floor(b * q + 0.5)
        w11 = np.floor(A * B / q + 0.5)
        w10, w01, w00 = A - w11, B - w11, q - A - B + w11
        return (w00 * img[j0, i0] + w10 * img[j0, i1] + w01 * img[j1, i0] + w11 * img[j1, i1]) / q
    return ((1 - a) * (1 - b) * img[j0, i0] + a * (1 - b) * img[j0, i1] + (1 - a) * b * img[j1, i0]
            + a * b * img[j1, i1])


def descriptors_f64(img, x, y, scale, ori_deg, frac_bits, want_flags=False):
    """ExtractSiftDescriptors_D, cuSIFT_D.cu:184-297, all keypoints at once, float64.
    want_flags: also return, per keypoint, whether any weighted sample's gradient points at 180 degrees to within
    rounding (dy ~ 0, dx < 0) -- there the algorithm is discontinuous (atan2f = +pi -> angle index 8 -> the NEXT
    cell's bin 0, :222-227; -pi + tiny -> index 0 -> the OWN cell's bin 0), so two precisions may legitimately differ."""
    img = img.astype(np.float64)
    n = len(x)
    tx = np.arange(16, dtype=np.float64)[None, None, :]   # [1, 1, 16]
    ty = np.arange(16, dtype=np.float64)[None, :, None]   # [1, 16, 1]
    gauss = np.exp(-(np.arange(16) - 7.5) ** 2 / 128.0)                      # :194
    theta = (2.0 * np.float64(np.float32(3.1415)) / 360.0) * ori_deg           # :199 (3.1415f is a float literal)
    sina, cosa = np.sin(theta)[:, None, None], np.cos(theta)[:, None, None]
    s = (12.0 / 16.0 * scale)[:, None, None]
    ssina, scosa = s * sina, s * cosa
    xpos = x[:, None, None] + (tx - 7.5) * scosa - (ty - 7.5) * ssina          # :207
    ypos = y[:, None, None] + (tx - 7.5) * ssina + (ty - 7.5) * scosa          # :208
    dx = tex2d_f64(img, xpos + cosa, ypos + sina, frac_bits) - tex2d_f64(img, xpos - cosa, ypos - sina, frac_bits)
    dy = tex2d_f64(img, xpos - sina, ypos + cosa, frac_bits) - tex2d_f64(img, xpos + sina, ypos - cosa, frac_bits)
    grad = gauss[None, :, None] * gauss[None, None, :] * np.sqrt(dx * dx + dy * dy)   # :213
    angf = 4.0 / np.float64(np.float32(3.1415)) * np.arctan2(dy, dx) + 4.0     # :214
    txi = np.arange(16)[None, None, :] + np.zeros((n, 16, 1), dtype=np.int64)
    tyi = np.arange(16)[None, :, None] + np.zeros((n, 1, 16), dtype=np.int64)
    hori = (txi + 2) // 4 - 1                                                  # :216
    horf = (txi - 1.5) / 4.0 - hori
    veri = (tyi + 2) // 4 - 1
    verf = (tyi - 1.5) / 4.0 - veri
    angi = angf.astype(np.int64)                                               # :222 (truncation; angf >= 0)
    angp = np.where(angi < 7, angi + 1, 0)
    fr = angf - angi
    hist = 8 * (4 * veri + hori)
    p1, p2 = angi + hist, angp + hist
    buf = np.zeros((n, 128), dtype=np.float64)
    kp = np.arange(n)[:, None, None] + np.zeros((1, 16, 16), dtype=np.int64)

    def add(idx, val, mask):
        ok = mask & (idx >= 0) & (idx < 128)   # indices outside `buffer` fall into `sums` and are overwritten (:259)
        np.add.at(buf, (kp[ok], idx[ok]), val[ok])

    left, right = txi >= 2, txi <= 14                                          # :229, :243 (sic: 14)
    upper, lower = tyi >= 2, tyi <= 13
    for hmask, hw, off_h in ((left, 1.0 - horf, 0), (right, horf, 8)):
        for vmask, vw, off_v in ((upper, 1.0 - verf, 0), (lower, verf, 32)):
            g2 = vw * (hw * grad)
            add(p1 + off_h + off_v, (1.0 - fr) * g2, hmask & vmask)
            add(p2 + off_h + off_v, fr * g2, hmask & vmask)
    buf = buf / np.sqrt((buf * buf).sum(axis=1, keepdims=True))                # :259-272
    buf = np.minimum(buf, 0.2)                                                 # :274-275
    out = buf / np.sqrt((buf * buf).sum(axis=1, keepdims=True))                # :277-291
    if want_flags:
        # index 8 is reached for angles in [3.1415f, pi]: a 9.3e-5 rad sliver below pi, plus rounding
        at_pi = (dx < 0) & (np.abs(dy) <= 2e-4 * np.abs(dx)) & (grad > 1e-9)
        return out, at_pi.any(axis=(1, 2))
    return out


@pytest.mark.parametrize("frac_bits", [0, 8])
def test_descriptor_restatement_vs_float64(oracle, gray1, frac_bits):
    """oracle_extract_descriptors on real keypoints == the float64 evaluation of cuSIFT_D.cu:184-297.

    What "equal" can mean: sample coordinates ~500 px carry a float32 rounding error of ~1.5e-5 px, the image has
    gradients of ~10..100 grey levels per px, so a float32 evaluation sits ~1e-5 L2 away from the float64 one -- that is
    the reference's own arithmetic noise, and the bar here.  Two discontinuities of the algorithm are exempt, because
    there two correct evaluations may differ: (1) a gradient at exactly 180 degrees (see descriptors_f64; flat or
    border-clamped neighbourhoods make dy exactly zero or +-1e-7), (2) with 8-bit fractions, a sample coordinate on a
    1/256 rounding step."""
    w, h = 640, 480
    src = pitched(gray1)
    dog = oracle.laplace_multi(src, w, h, 0.0)
    pts, n = oracle.find_points_multi(dog, w, h, 0.5, 10.0, 1.0, 16384)
    assert n > 1500
    oracle.compute_orientations(src, w, h, pts, 0, n, frac_bits)
    fin = np.isfinite(pts["orientation"][:n])
    before = pts[:n].copy()
    oracle.extract_descriptors(src, w, h, pts, 0, n, 1.0, frac_bits)
    want, at_pi = descriptors_f64(gray1, before["coords2D"][:, 0].astype(np.float64),
                                  before["coords2D"][:, 1].astype(np.float64), before["scale"].astype(np.float64),
                                  before["orientation"].astype(np.float64), frac_bits, want_flags=True)
    got = pts["data"][:n].astype(np.float64)
    ok = fin & np.isfinite(want).all(axis=1)
    assert ok.mean() > 0.999
    l2 = np.linalg.norm(got - want, axis=1)
    smooth = ok & ~at_pi
    assert smooth.mean() > 0.5
    if frac_bits == 0:
        assert np.median(l2[smooth]) < 3e-5
        assert l2[smooth].max() < 3e-4, np.sort(l2[smooth])[-5:]
    else:
        # each descriptor makes 2048 fraction roundings; a 1.5e-5 px float32 coordinate error against a 1/256 px step
        # flips ~0.4 % of them (~8 per descriptor, each worth 1/256 of a local gradient): ~1e-4 L2 is the model's own
        # sensitivity to float32 coordinates, so this leg anchors the quantisation MODEL, the other one the arithmetic
        # (round 6: the weights are 8-bit fixed point too -- a flipped fraction moves a tap by up to 1/256 of a pixel
        # difference instead of 1/65536 steps of the products.  Two keypoints are allowed to combine both discontinuities:
        # the flipped fraction makes dy of a 180-degree sample exactly zero in one evaluation only, which the float64
        # flag cannot see -- one such keypoint exists on the fixture, 0.059 away; everything else is below 7e-4)
        top = np.sort(l2[smooth])
        assert np.median(top) < 5e-4 and top[-3] < 3e-3 and top[-1] < 0.1, (np.median(top), top[-5:])
    # at the 180-degree discontinuity the two evaluations still describe the same patch
    assert l2[ok].max() < 0.2
    np.testing.assert_allclose(np.linalg.norm(got[ok], axis=1), 1.0, atol=1e-5)


def test_descriptor_quirk_paths_vs_float64(oracle):
    """The reference's index-overflow paths in the float64 evaluation and in the restatement: the
    `tx<=14` spill into the next row's first cell, and angle index 8 (atan2f == +pi: identical rows, orientation 0,
    falling ramp) spilling into the next cell's bin 0."""
    w, h = 256, 192
    x = np.arange(w, dtype=np.float32)
    row = np.where((x // 16) % 2 == 0, 200 - 8 * (x % 16), 72 + 8 * (x % 16))
    img = np.tile(row, (h, 1)).astype(np.float32)
    from oracle_binding import SIFT_POINT_DTYPE

    pts = np.zeros(24, dtype=SIFT_POINT_DTYPE)
    rng = np.random.default_rng(9)
    pts["coords2D"][:, 0] = rng.uniform(20, w - 20, 24)
    pts["coords2D"][:, 1] = rng.uniform(20, h - 20, 24)
    pts["scale"] = rng.uniform(0.9, 2.5, 24)
    pts["orientation"] = np.where(np.arange(24) % 2 == 0, 0.0, rng.uniform(0, 360, 24))
    before = pts.copy()
    # 8-bit fractions: weights are multiples of 2^-16 and the image is integer-valued, so every tap is exact in both
    # precisions and dy == +0 exactly where the reference would see it (with exact fractions dy = +-1e-7 at random)
    oracle.extract_descriptors(pitched(img), w, h, pts, 0, 24, 1.0, 8)
    want, at_pi = descriptors_f64(img, before["coords2D"][:, 0].astype(np.float64),
                                  before["coords2D"][:, 1].astype(np.float64), before["scale"].astype(np.float64),
                                  before["orientation"].astype(np.float64), 8, want_flags=True)
    assert at_pi[::2].all()  # the orientation-0 keypoints do hit the angle-index-8 path
    l2 = np.linalg.norm(pts["data"].astype(np.float64) - want, axis=1)
    assert l2[::2].max() < 1e-5, l2   # orientation 0: no fraction step can flip (rows identical, columns on a lattice)
    assert l2.max() < 5e-3, l2


# ------------------------------------------------------------------------------------------------
# extrema + refinement, cuSIFT_D.cu:402-523, float64 on the oracle's float32 DoG planes
# ------------------------------------------------------------------------------------------------
def find_points_f64(dog, w, h, thresh, edge_limit):
    """All (x, y, s) that pass the 26-neighbour test (float32 comparisons are exact in float64), then the edge test
    and the 3-D quadratic refinement in float64.  Returns dict of arrays incl. the branch taken."""
    D = dog[:, :h, :w].astype(np.float64)
    out = {k: [] for k in ("x", "y", "scale", "sharp", "edge", "fallback", "margin")}
    for s in range(5):                                            # NUM_SCALES, centre plane s + 1
        c = D[s + 1, 1:-1, 1:-1]
        nb = []
        for p in (s, s + 1, s + 2):
            for dy in (0, 1, 2):
                for dx in (0, 1, 2):
                    if p == s + 1 and dy == 1 and dx == 1:
                        continue
                    nb.append(D[p, dy:dy + h - 2, dx:dx + w - 2])
        nb = np.stack(nb)
        is_ext = ((c < -thresh) & (c < nb.min(axis=0))) | ((c > thresh) & (c > nb.max(axis=0)))   # :451-470
        ys, xs = np.nonzero(is_ext)
        ys, xs = ys + 1, xs + 1
        P, C, Q = D[s], D[s + 1], D[s + 2]
        v = C[ys, xs]
        dxx = 2 * v - C[ys, xs - 1] - C[ys, xs + 1]               # :479
        dyy = 2 * v - C[ys - 1, xs] - C[ys + 1, xs]
        dxy = 0.25 * (C[ys + 1, xs + 1] + C[ys - 1, xs - 1] - C[ys - 1, xs + 1] - C[ys + 1, xs - 1])
        tra, det = dxx + dyy, dxx * dyy - dxy * dxy
        keep = tra * tra < edge_limit * det                       # :486 (edgeThresh used raw)
        dx = 0.5 * (C[ys, xs + 1] - C[ys, xs - 1])
        dy = 0.5 * (C[ys + 1, xs] - C[ys - 1, xs])
        ds = 0.5 * (P[ys, xs] - Q[ys, xs])                        # :493
        dss = 2 * v - Q[ys, xs] - P[ys, xs]
        dxs = 0.25 * (Q[ys, xs + 1] + P[ys, xs - 1] - P[ys, xs + 1] - Q[ys, xs - 1])
        dys = 0.25 * (Q[ys + 1, xs] + P[ys - 1, xs] - Q[ys - 1, xs] - P[ys + 1, xs])
        idxx, idxy, idxs = dyy * dss - dys * dys, dys * dxs - dxy * dss, dxy * dys - dyy * dxs
        with np.errstate(all="ignore"):
            idet = 1.0 / (idxx * dxx + idxy * dxy + idxs * dxs)
            idyy, idys, idss = dxx * dss - dxs * dxs, dxy * dxs - dxx * dys, dxx * dyy - dxy * dxy
            pdx = idet * (idxx * dx + idxy * dy + idxs * ds)
            pdy = idet * (idxy * dx + idyy * dy + idys * ds)
            pds = idet * (idxs * dx + idys * dy + idss * ds)
            big = np.maximum(np.maximum(np.abs(pdx), np.abs(pdy)), np.abs(pds))
            fb = ~(big <= 0.5)                                    # :503-504
            pdx, pdy, pds = np.where(fb, dx / dxx, pdx), np.where(fb, dy / dyy, pdy), np.where(fb, ds / dss, pds)
            sharp = v + 0.5 * (dx * pdx + dy * pdy + ds * pds)
            edge = tra * tra / det
        k = keep
        out["x"].append((xs + pdx)[k]); out["y"].append((ys + pdy)[k])
        out["scale"].append((2.0 ** (s / 5.0) * 2.0 ** (pds / 5.0))[k])   # d_Scales[s] * exp2f(pds * factor), :507
        out["sharp"].append(sharp[k]); out["edge"].append(edge[k]); out["fallback"].append(fb[k])
        out["margin"].append(np.abs(big - 0.5)[k])
    return {k: np.concatenate(v) for k, v in out.items()}


@pytest.mark.parametrize("init_blur,thresh", [(0.0, 0.5), (0.8, 0.1)])
def test_refinement_restatement_vs_float64(oracle, gray1, init_blur, thresh):
    """Location, scale, sharpness and edgeness of oracle_find_points_multi == float64 evaluation of
    cuSIFT_D.cu:474-521 on the same DoG planes (the golden file pins location and scale only, and only for
    initBlur = 0)."""
    w, h = 640, 480
    dog = oracle.laplace_multi(pitched(gray1), w, h, init_blur)
    pts, n = oracle.find_points_multi(dog, w, h, thresh, 10.0, 1.0, 32768)
    ref = find_points_f64(dog, w, h, thresh, 10.0)
    assert n > 500
    # the edge test `tra^2 < 10 det` may fall either way in float32 for a few borderline points
    assert abs(n - len(ref["x"])) <= max(3, n // 500), (n, len(ref["x"]))
    mine = np.stack([pts["coords2D"][:n, 0], pts["coords2D"][:n, 1], pts["scale"][:n]], axis=1).astype(np.float64)
    theirs = np.stack([ref["x"], ref["y"], ref["scale"]], axis=1)
    idx, dist = match_nearest(theirs, mine, 1e-2)
    # well-conditioned points (the |p| <= 0.5 decision is not marginal, the solve is not the fallback's)
    good = (ref["margin"] > 1e-3) & (dist < 0.5)
    assert good.mean() > 0.97
    assert np.median(dist[good]) < 1e-5 and (dist[good] < 1e-3).mean() > 0.995
    sh, ed = pts["sharpness"][idx].astype(np.float64), pts["edgeness"][idx].astype(np.float64)
    tight = good & (dist < 1e-3)
    np.testing.assert_allclose(sh[tight], ref["sharp"][tight], rtol=2e-4, atol=1e-4)
    # edgeness = tra^2/det amplifies the float32 rounding of det; relative agreement in the bulk
    rel = np.abs(ed[tight] - ref["edge"][tight]) / np.abs(ref["edge"][tight])
    assert np.median(rel) < 1e-5 and (rel < 1e-2).mean() > 0.99


# ------------------------------------------------------------------------------------------------
# LaplaceMulti with a non-zero initBlur, cuSIFT.cu:399-422 + cuSIFT_D.cu:525-553
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("init_blur", [0.0, 0.5, 0.8, 1.0])
def test_laplace_restatement_vs_float64(oracle, gray1, init_blur):
    """Tap table and DoG planes against a float64 evaluation: level i has sigma_i = 2^((i-1)/5), variance
    sigma_i^2 - initBlur^2, 9 taps exp(-j^2 / (2 var)) normalised; vertical then horizontal, clamped borders.
    (var <= 1e-6 => identity taps is the documented rule of this build; the reference yields NaN there.)"""
    w, h = 640, 480
    taps = oracle.laplace_taps(init_blur).reshape(8, 16)[:, :9].astype(np.float64)
    img = gray1.astype(np.float64)
    L = []
    for i in range(8):
        var = (2.0 ** ((i - 1) / 5.0)) ** 2 - init_blur ** 2
        if var <= 1e-6:
            k = np.zeros(9)
            k[4] = 1.0
        else:
            k = np.exp(-np.arange(-4, 5) ** 2 / (2.0 * var))
            k /= k.sum()
        np.testing.assert_allclose(taps[i], k, rtol=2e-6, atol=1e-9)
        pad = np.pad(img, ((4, 4), (0, 0)), mode="edge")
        v = sum(k[j] * pad[j:j + h, :] for j in range(9))
        pad = np.pad(v, ((0, 0), (4, 4)), mode="edge")
        L.append(sum(k[j] * pad[:, j:j + w] for j in range(9)))
    want = np.stack([L[s] - L[s + 1] for s in range(7)])
    got = oracle.laplace_multi(pitched(gray1), w, h, init_blur)[:, :, :w].astype(np.float64)
    assert np.abs(got - want).max() < 2e-4   # values ~1e2, float32 9+9-term sums
