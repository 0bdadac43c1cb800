"""MatchSiftData: oracle pinned by the reference's fixtures; HIP matcher vs oracle."""
import os

import numpy as np
import pytest

from oracle_binding import SIFT_POINT_DTYPE, read_match_indices, read_vlfeat_sift

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def vl_pair():
    return (read_vlfeat_sift(os.path.join(GOLDEN, "vlfeat_sift1.bin")),
            read_vlfeat_sift(os.path.join(GOLDEN, "vlfeat_sift2.bin")))


# ---- oracle vs the reference's golden vectors (test/test.cpp:25-56) ----
def test_oracle_reproduces_matlab_match_indices(oracle, vl_pair):
    s1, s2 = vl_pair[0].copy(), vl_pair[1]
    assert len(s1) == 884 and len(s2) == 856
    oracle.match(s1, s2, 1)
    ii, jj = read_match_indices(os.path.join(GOLDEN, "match_indices1_2.bin"))
    assert len(ii) == 326
    # EXPECT_EQ(indices_j[i], matches[indices_i[i] - 1]->pt1->match + 1)
    np.testing.assert_array_equal(s1["match"][ii - 1] + 1, jj)


def test_oracle_ratio_test_gives_340_matches(oracle, vl_pair):
    s1, s2 = vl_pair[0].copy(), vl_pair[1]
    oracle.match(s1, s2, 1)
    assert len(oracle.match_filter(s1, 1000.0, 0.6)) == 340  # EXPECT_EQ(340, matches.size())
    assert len(oracle.match_filter(s1)) == 884              # default thresholds keep every point


def test_oracle_scores_are_true_top2(oracle, vl_pair):
    s1, s2 = vl_pair[0].copy(), vl_pair[1]
    oracle.match(s1, s2, 1)
    d = 2.0 - 2.0 * (s1["data"].astype(np.float64) @ s2["data"].astype(np.float64).T)
    srt = np.sort(d, axis=1)
    np.testing.assert_allclose(s1["score"], srt[:, 0], atol=2e-6)
    # score = 2 - 2*dot cancels for near-duplicate descriptors: absolute error ~2e-7 on score, hence atol on the ratio
    np.testing.assert_allclose(s1["ambiguity"], srt[:, 0] / (srt[:, 1] + 1e-6), rtol=1e-4, atol=3e-5)
    assert (s1["match"] == d.argmin(axis=1)).mean() > 0.999
    np.testing.assert_array_equal(s1["match_xpos"], s2["coords2D"][s1["match"], 0])
    # dot-product mode
    t1 = vl_pair[0].copy()
    oracle.match(t1, s2, 0)
    dot = s1["data"].astype(np.float64) @ s2["data"].astype(np.float64).T
    np.testing.assert_allclose(t1["score"], dot.max(axis=1), atol=2e-6)


# ---- HIP matcher ----
def gpu_match(ctx, s1, s2, distance):
    from cusift_amd.capi import DeviceBuffer

    d1 = DeviceBuffer.from_numpy(ctx, s1)
    d2 = DeviceBuffer.from_numpy(ctx, s2)
    ctx.match(d1.ptr, len(s1), d2.ptr, len(s2), distance)
    ctx.synchronize()
    return d1.to_numpy(SIFT_POINT_DTYPE, (len(s1),))


@pytest.mark.gpu
@pytest.mark.parametrize("distance", [1, 0])
def test_gpu_matcher_vs_oracle_on_reference_fixture(ctx, oracle, vl_pair, distance):
    s1, s2 = vl_pair
    want = s1.copy()
    oracle.match(want, s2, distance)
    got = gpu_match(ctx, s1, s2, distance)
    # different (but fixed) summation order of the 128-term dot product: scores agree to ~1e-7
    np.testing.assert_allclose(got["score"], want["score"], atol=2e-6, rtol=0)
    same = got["match"] == want["match"]
    assert same.mean() >= 0.998, same.mean()
    np.testing.assert_allclose(got["ambiguity"][same], want["ambiguity"][same], rtol=1e-4, atol=3e-5)
    np.testing.assert_array_equal(got["match_xpos"], s2["coords2D"][got["match"], 0])
    np.testing.assert_array_equal(got["match_ypos"], s2["coords2D"][got["match"], 1])
    # untouched fields
    np.testing.assert_array_equal(got["data"], s1["data"])
    np.testing.assert_array_equal(got["coords2D"], s1["coords2D"])


@pytest.mark.gpu
def test_gpu_matcher_passes_the_reference_tests(ctx, vl_pair):
    """test/test.cpp:25-56 on the HIP path itself: 326 MATLAB pairs and 340 ratio-test matches."""
    from cusift_amd import capi

    s1, s2 = vl_pair
    got = gpu_match(ctx, s1, s2, 1)
    ii, jj = read_match_indices(os.path.join(GOLDEN, "match_indices1_2.bin"))
    np.testing.assert_array_equal(got["match"][ii - 1] + 1, jj)
    assert len(capi.match_filter(got, 1000.0, 0.6)) == 340


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2", [(1, 1), (1, 40), (17, 16), (64, 33), (65, 257), (300, 7), (1000, 999)])
def test_gpu_matcher_ragged_sizes(ctx, oracle, n1, n2):
    rng = np.random.default_rng(n1 * 1000 + n2)

    def rand_pts(n):
        p = np.zeros(n, dtype=SIFT_POINT_DTYPE)
        d = np.abs(rng.normal(size=(n, 128))).astype(np.float32)
        p["data"] = d / np.linalg.norm(d, axis=1, keepdims=True)
        p["coords2D"] = rng.uniform(0, 1000, (n, 2)).astype(np.float32)
        return p

    s1, s2 = rand_pts(n1), rand_pts(n2)
    for distance in (1, 0):
        want = s1.copy()
        oracle.match(want, s2, distance)
        got = gpu_match(ctx, s1, s2, distance)
        np.testing.assert_allclose(got["score"], want["score"], atol=2e-6, rtol=0)
        same = got["match"] == want["match"]
        assert same.mean() >= 0.99
        np.testing.assert_allclose(got["ambiguity"][same], want["ambiguity"][same], rtol=1e-4, atol=3e-5)
        assert (got["match"] >= 0).all() and (got["match"] < n2).all()


@pytest.mark.gpu
def test_gpu_matcher_signed_descriptors_hit_the_l2_clamp(ctx, oracle):
    """ComputeL2Distance writes FLT_MAX (999) where the dot product is <= -1 (extras/matching.cu:71-72).  Signed unit
    vectors and their exact negatives reach that branch; a row whose every column is clamped keeps match = -1."""
    rng = np.random.default_rng(41)
    n1, n2 = 70, 95
    d = rng.normal(size=(n1, 128)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    s1 = np.zeros(n1, dtype=SIFT_POINT_DTYPE)
    s1["data"] = d
    s2 = np.zeros(n2, dtype=SIFT_POINT_DTYPE)
    e = rng.normal(size=(n2, 128)).astype(np.float32)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    e[:n1:2] = -d[::2][: len(e[:n1:2])] * np.float32(1.5)  # dot = -1.5 with the row of the same index
    s2["data"] = e
    s2["coords2D"] = rng.uniform(0, 500, (n2, 2)).astype(np.float32)
    for distance in (1, 0):
        want = s1.copy()
        oracle.match(want, s2, distance)
        got = gpu_match(ctx, s1, s2, distance)
        np.testing.assert_allclose(got["score"], want["score"], atol=4e-6, rtol=0)
        np.testing.assert_array_equal(got["match"], want["match"])
    # every column clamped: one row, all of image 2 = -2 x that row
    one = s1[:1].copy()
    allneg = np.zeros(40, dtype=SIFT_POINT_DTYPE)
    allneg["data"] = -2.0 * one["data"][0]
    want = one.copy()
    oracle.match(want, allneg, 1)
    got = gpu_match(ctx, one, allneg, 1)
    assert got["match"][0] == want["match"][0] == -1
    assert got["score"][0] == want["score"][0] == np.float32(999.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2", [(3000, 2900), (200, 5000), (5000, 130)])
def test_gpu_matcher_column_splits_fold_to_the_single_scan(ctx, oracle, n1, n2):
    """The matcher splits image 2's columns over the grid (auto: several splits at these sizes) and folds the splits
    in column order; forced to one split it is the plain scan.  Both must agree with each other and the oracle."""
    from cusift_amd import capi

    rng = np.random.default_rng(n1 + n2)

    def rand_pts(n):
        p = np.zeros(n, dtype=SIFT_POINT_DTYPE)
        d = np.abs(rng.normal(size=(n, 128))).astype(np.float32)
        p["data"] = d / np.linalg.norm(d, axis=1, keepdims=True)
        p["coords2D"] = rng.uniform(0, 1000, (n, 2)).astype(np.float32)
        return p

    s1, s2 = rand_pts(n1), rand_pts(n2)
    for distance in (1, 0):
        runs = {}
        for splits in ("auto", "1", "7", "1000"):
            ctx.set_policy(capi.POLICY_MATCH_SPLITS, 0 if splits == "auto" else int(splits))
            runs[splits] = gpu_match(ctx, s1, s2, distance)
        ctx.set_policy(capi.POLICY_MATCH_SPLITS, 0)
        for k in ("auto", "7", "1000"):
            for f in ("score", "ambiguity", "match", "match_xpos", "match_ypos"):
                np.testing.assert_array_equal(runs[k][f], runs["1"][f], err_msg="%s splits, %s" % (k, f))
        want = s1.copy()
        oracle.match(want, s2, distance)
        np.testing.assert_allclose(runs["auto"]["score"], want["score"], atol=2e-6, rtol=0)
        assert (runs["auto"]["match"] == want["match"]).mean() >= 0.99
        np.testing.assert_array_equal(runs["auto"]["match_xpos"], s2["coords2D"][runs["auto"]["match"], 0])


@pytest.mark.gpu
def test_gpu_matcher_on_extracted_siftdata(ctx, oracle, gray1):
    """End to end: extract two views on the GPU, match them on the GPU, compare with the oracle's matcher."""
    from cusift_amd import capi
    from cusift_amd.capi import DeviceBuffer

    prm = capi.default_params(num_octaves=4, init_blur=0.0, peak_thresh=1.5, max_pts=4096)
    imgs = [gray1, np.roll(gray1, (3, 5), axis=(0, 1))]
    bufs, hosts = [], []
    for im in imgs:
        d = DeviceBuffer(ctx, prm.max_pts * 588)
        h = np.zeros(prm.max_pts, dtype=SIFT_POINT_DTYPE)
        n = ctx.extract_host(im, prm, d.ptr, h)
        bufs.append((d, n))
        hosts.append(h[:n].copy())
    (d1, n1), (d2, n2) = bufs
    assert n1 > 300 and n2 > 300
    ctx.match(d1.ptr, n1, d2.ptr, n2, 1)
    ctx.synchronize()
    got = d1.to_numpy(SIFT_POINT_DTYPE, (n1,))
    want = hosts[0].copy()
    oracle.match(want, hosts[1], 1)
    np.testing.assert_allclose(got["score"], want["score"], atol=5e-6, rtol=0)
    assert (got["match"] == want["match"]).mean() > 0.995
    good = capi.match_filter(got, 1000.0, 0.6)
    assert len(good) > 100
    # the second view is the first shifted by (5, 3) px: good matches must agree with that shift
    dx = got["match_xpos"][good] - got["coords2D"][good, 0]
    dy = got["match_ypos"][good] - got["coords2D"][good, 1]
    assert np.median(np.abs(dx - 5.0)) < 0.05 and np.median(np.abs(dy - 3.0)) < 0.05
