"""Pins the oracle against the reference's own golden vectors.

Input  tests/golden/gray1.pgm        (= test/data/gray1, 640x480)
Output tests/golden/cusift1_check.bin and cusift1.bin (= two runs of the reference, 4096 rows of
       x, y, scale, orientation) with the parameters of test/detector.cpp:37-49.

The golden run saturated its 4096-point buffer: file rows 0..1554 are ALL points of octaves 5..1
(7+20+87+261+1180) and rows 1555..4095 are a racy subset of octave 0's 7953 points, so octaves >= 1
are compared strictly and octave 0 as "every golden row is one of ours".
"""
import numpy as np
import pytest

from parity_utils import ang_diff, canonical_order, golden_gates, match_nearest, orientation_outliers, xys


@pytest.fixture(scope="module", params=["", "libm"], ids=["shared-math", "glibc"])
def oracle(request):
    """Every gate of this module is applied to BOTH builds of the restatement: the one whose device-side
    transcendental functions are the written-out ones it shares with the HIP kernels (cusift_amd/csrc/sift_math.h)
    and the one on glibc's expf/exp2f/atan2f/sinf/cosf.  The reference ran on CUDA's libm, which is neither."""
    from oracle_binding import Oracle

    return Oracle(request.param)

REF_PARAMS = dict(num_octaves=6, init_blur=0.0, peak_thresh=0.1, edge_thresh=10.0, lowest_scale=0.0,
                  subsampling=1.0, max_pts=16384)
N_COARSE = 1555  # golden rows 0..1554 = octaves 5..1
COARSE_COUNTS = {32.0: 7, 16.0: 20, 8.0: 87, 4.0: 261, 2.0: 1180}


def test_oracle_counts_match_golden_layout(oracle, gray1):
    pts = oracle.extract(gray1, **REF_PARAMS)
    sub = pts["subsampling"]
    for s, n in COARSE_COUNTS.items():
        assert int((sub == s).sum()) == n, (s, int((sub == s).sum()), n)
    assert int((sub == 1.0).sum()) == 7953
    # the reference emits octave blocks coarsest first (cuSIFT.cu:190-196)
    assert np.all(np.diff(sub) <= 0)


def test_oracle_location_scale_vs_golden(oracle, gray1, golden_check):
    pts = oracle.extract(gray1, **REF_PARAMS)
    mine = xys(pts)
    gold = golden_check.astype(np.float64)
    idx, dist = match_nearest(gold[:N_COARSE, :3], mine, 1e-2)
    # octaves >= 1: every golden row found within 1e-2 (base pixels), >= 98 % within 1e-3
    assert (dist < 1e-2).all(), int((dist >= 1e-2).sum())
    assert (dist < 1e-3).mean() >= 0.98, (dist < 1e-3).mean()
    assert len(set(idx.tolist())) == N_COARSE  # one-to-one
    # matched points are in coarse octaves
    assert (pts["subsampling"][idx] >= 2.0).all()
    # octave 0: the golden subset (racy at the 4096 cap) is contained in ours
    idx0, dist0 = match_nearest(gold[N_COARSE:, :3], mine, 1e-2)
    assert (dist0 < 1e-2).mean() >= 0.995, (dist0 < 1e-2).mean()


def test_oracle_orientation_vs_golden(oracle, gray1, golden_check):
    pts = oracle.extract(gray1, **REF_PARAMS)
    gold = golden_check.astype(np.float64)
    idx, dist = match_nearest(gold[:N_COARSE, :3], xys(pts), 1e-2)
    d = ang_diff(gold[:N_COARSE, 3], pts["orientation"][idx].astype(np.float64))
    # the north star's 1e-3 (degrees) against the reference's OWN run, since the texture model's weights are fixed point
    # too (oracle_tex2d; SURVEY.md hard part 1 expected no better than 94 % within 0.1 degree)
    assert (d < 1e-3).mean() >= 0.99, (d < 1e-3).mean()
    assert (d < 0.1).mean() >= 0.995, (d < 0.1).mean()
    assert (d < 1.0).mean() >= 0.999, (d < 1.0).mean()
    assert np.median(d) < 1e-4


@pytest.mark.parametrize("which", ["cusift1_check", "cusift1"])
def test_oracle_vs_all_of_the_golden_file(oracle, gray1, golden_check, golden_run2, which, record_property):
    """Everything the reference's two runs hold -- 2 x 4096 rows of x, y, scale AND orientation -- against the oracle:
    the octave-0 rows (1555..4095) are gated on orientation too, and `cusift1` (the second run) gets the same gates as
    `cusift1_check`, not only a location check.  The achieved fractions are in the assertion message / test properties."""
    pts, peaks = oracle.extract_with_orientation_peaks(gray1, **REF_PARAMS)
    gold = golden_check if which == "cusift1_check" else golden_run2
    got = golden_gates(gold, pts, "oracle")
    for k, v in got.items():
        record_property(k, v)
    # the orientation tail, explained: every golden row more than 1 degree from ours must sit at the oracle's SECOND
    # histogram peak with the two peaks nearly tied (the reference computes both and keeps the larger, cuSIFT_D.cu:362-394:
    # a tie decided the other way by the order of its LDS atomics) -- one row of 4,095 on this fixture, peak ratio 0.983
    tail = orientation_outliers(gold, pts, peaks)
    for k, v in tail.items():
        record_property("ori_tail_" + k, str(v))
    assert tail["outliers_gt_1_deg"] <= 3 and tail["outliers_unexplained"] == 0, tail
    assert tail["outliers_within_a_bin_of_first_peak"] == 0, tail
    assert tail["outliers_at_second_peak"] == tail["outliers_at_second_peak_ratio_gt_0.90"], tail


def test_texture_weights_are_fixed_point(oracle, gray1, golden_check):
    """The rule itself, at the GOLDEN locations (so that no difference in x, y or scale is in the way): orientations
    recomputed by the oracle's stage function at the golden x, y, scale of every matched row equal the golden
    orientations to 1e-4 degree -- all of them -- and most are the same float."""
    from oracle_binding import SIFT_POINT_DTYPE, pitched

    pts = oracle.extract(gray1, **REF_PARAMS)
    gold = golden_check.astype(np.float64)
    idx, dist = match_nearest(gold[:, :3], xys(pts), 1e-2)
    ok = dist < 1e-2
    assert ok.sum() >= 4090
    sub = pts["subsampling"][idx]
    h, w = gray1.shape
    imgs, dims = [pitched(gray1)], [(w, h)]
    for _ in range(5):
        ww, hh = dims[-1]
        imgs.append(oracle.scale_down(imgs[-1], ww, hh))
        dims.append((ww // 2, hh // 2))
    re = np.zeros(len(gold), dtype=SIFT_POINT_DTYPE)
    re["coords2D"] = (golden_check[:, :2] / sub[:, None]).astype(np.float32)  # exact: powers of two
    re["scale"] = (golden_check[:, 2] / sub).astype(np.float32)
    for k in range(6):
        sel = np.flatnonzero(sub == 2.0 ** k)
        tmp = re[sel].copy()
        oracle.compute_orientations(imgs[k], dims[k][0], dims[k][1], tmp, 0, len(tmp))
        re["orientation"][sel] = tmp["orientation"]
    d = ang_diff(gold[ok, 3], re["orientation"][ok].astype(np.float64))
    assert d.max() < 1e-4, (d.max(), int((d >= 1e-4).sum()))
    assert (re["orientation"][ok] == golden_check[ok, 3]).mean() > 0.8  # bit-identical floats


def test_two_reference_runs_differ_only_in_octave0_order(golden_check, golden_run2):
    """What the two golden files say about the reference itself: the coarse rows are the same points (append order
    inside an octave is racy), the octave-0 rows two different racy subsets of the same 7953."""
    a, b = golden_check.astype(np.float64), golden_run2.astype(np.float64)
    idx, dist = match_nearest(a[:N_COARSE, :3], b[:N_COARSE, :3], 1e-2)
    assert (dist == 0.0).all() and len(set(idx.tolist())) == N_COARSE
    assert (ang_diff(a[:N_COARSE, 3], b[idx, 3]) < 1e-3).mean() > 0.99  # LDS float atomics: run-to-run ulp noise


def test_texture_model_fraction_bits(oracle, gray1, golden_check):
    """The 8-bit-fraction texture model tracks the golden orientations better than exact fp32."""
    gold = golden_check.astype(np.float64)
    med = {}
    for bits in (8, 0):
        pts = oracle.extract(gray1, tex_frac_bits=bits, **REF_PARAMS)
        idx, _ = match_nearest(gold[:N_COARSE, :3], xys(pts), 1e-2)
        med[bits] = np.median(ang_diff(gold[:N_COARSE, 3], pts["orientation"][idx].astype(np.float64)))
    assert med[8] < med[0]


def test_scale_down_vertical_taps_are_asymmetric(oracle):
    """cuSIFT_D.cu:75,123-125: rows (2r-1,2r,2r+1,2r+2,2r+3) with weights (k1,k2,k1,k0,k0)."""
    from oracle_binding import pitched

    h, w = 16, 8
    k = np.exp(-np.arange(-2, 3, dtype=np.float64) ** 2 / 2.0 / 0.5).astype(np.float32)
    k /= k.sum(dtype=np.float32)
    for row, expect in ((5, k[1]), (6, k[2]), (7, k[1]), (8, k[0]), (9, k[0]), (4, 0.0), (10, 0.0)):
        img = np.zeros((h, w), dtype=np.float32)
        img[row, :] = 1.0
        out = oracle.scale_down(pitched(img), w, h)
        np.testing.assert_allclose(out[3, 1], expect, rtol=1e-6, atol=1e-7)


def test_laplace_taps_degenerate_rule(oracle):
    """initBlur >= level sigma: identity taps (documented rule; the reference yields NaN there)."""
    t = oracle.laplace_taps(1.0).reshape(8, 16)
    assert np.isfinite(t).all()
    np.testing.assert_array_equal(t[0, :9], np.eye(9, dtype=np.float32)[4])
    np.testing.assert_array_equal(t[1, :9], np.eye(9, dtype=np.float32)[4])
    assert t[2, 4] < 1.0 and abs(t[2, :9].sum() - 1.0) < 1e-6
    t0 = oracle.laplace_taps(0.0).reshape(8, 16)
    np.testing.assert_allclose(t0[:, :9].sum(axis=1), 1.0, atol=1e-6)


def test_oracle_point_record_layout():
    from oracle_binding import SIFT_POINT_DTYPE

    assert SIFT_POINT_DTYPE.itemsize == 588
    offs = {n: SIFT_POINT_DTYPE.fields[n][1] for n in SIFT_POINT_DTYPE.names}
    assert offs["coords2D"] == 0 and offs["scale"] == 8 and offs["orientation"] == 20
    assert offs["subsampling"] == 48 and offs["data"] == 64 and offs["coords3D"] == 576


def test_oracle_overflow_and_lowest_scale(oracle, gray1):
    prm = dict(REF_PARAMS)
    prm["max_pts"] = 100
    pts = oracle.extract(gray1, **prm)
    assert len(pts) == 100  # numPts = min(counter, maxPts), cuSIFT.cu:110
    prm = dict(REF_PARAMS)
    prm["lowest_scale"] = 2.0  # octave 0 (subsampling 1) is skipped: 2.0 < 1*2 is false (cuSIFT.cu:194)
    pts = oracle.extract(gray1, **prm)
    assert (pts["subsampling"] >= 2.0).all() and len(pts) == 1555


def test_shared_math_build_agrees_with_glibc_build(gray1):
    """The two builds differ only in the last bits of five functions: same keypoints, same locations, scales within a
    few ulp, and -- because a last bit occasionally decides a 1/256 texture-fraction step or a histogram bin --
    orientations and descriptors equal within the north-star tolerances for >= 99 % of the points."""
    from oracle_binding import Oracle

    a = canonical_order(Oracle("").extract(gray1, **REF_PARAMS))
    b = canonical_order(Oracle("libm").extract(gray1, **REF_PARAMS))
    assert len(a) == len(b) == 9508
    np.testing.assert_array_equal(a["sharpness"], b["sharpness"])
    np.testing.assert_array_equal(a["edgeness"], b["edgeness"])
    sub = a["subsampling"].astype(np.float64)
    assert (np.abs(a["coords2D"].astype(np.float64) - b["coords2D"]).max(axis=1) / sub).max() == 0.0
    np.testing.assert_allclose(a["scale"], b["scale"], rtol=1e-6)
    d = ang_diff(a["orientation"].astype(np.float64), b["orientation"].astype(np.float64))
    fin = np.isfinite(d)
    assert fin.mean() > 0.999 and (d[fin] < 1e-3).mean() >= 0.99
    l2 = np.linalg.norm(a["data"][fin].astype(np.float64) - b["data"][fin], axis=1)
    assert (l2 < 1e-4).mean() >= 0.98 and np.median(l2) < 1e-6
