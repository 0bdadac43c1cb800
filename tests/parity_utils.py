"""Shared helpers of the parity tests (set-based comparison of SiftPoint arrays)."""
import numpy as np


def canonical_order(pts):
    """Deterministic order: octave (coarse first, as the reference emits them), then y, x, scale."""
    key = np.lexsort((pts["scale"], pts["coords2D"][:, 0], pts["coords2D"][:, 1], -pts["subsampling"]))
    return pts[key]


def ang_diff(a, b):
    d = np.abs(a - b) % 360.0
    return np.minimum(d, 360.0 - d)


def match_nearest(ref_xys, got_xys, tol):
    """For every row of ref (x,y,scale) the index of the nearest row of got and the max-abs distance."""
    idx = np.empty(len(ref_xys), dtype=np.int64)
    dist = np.empty(len(ref_xys), dtype=np.float64)
    # chunked brute force (a few thousand points)
    for s in range(0, len(ref_xys), 512):
        blk = ref_xys[s:s + 512]
        d = np.abs(blk[:, None, :] - got_xys[None, :, :]).max(axis=2)
        j = d.argmin(axis=1)
        idx[s:s + 512] = j
        dist[s:s + 512] = d[np.arange(len(blk)), j]
    return idx, dist


def xys(pts):
    return np.stack([pts["coords2D"][:, 0], pts["coords2D"][:, 1], pts["scale"]], axis=1).astype(np.float64)


N_COARSE = 1555  # golden rows 0..1554 = octaves 5..1 (7 + 20 + 87 + 261 + 1180); rows 1555..4095 = a subset of octave 0


def golden_gates(gold, pts, label):
    """Every gate SURVEY.md section 8c recommends, over EVERYTHING a golden file of the reference holds (4096 rows of
    x, y, scale, orientation; test/detector.cpp:65-84 compares the same four fields): location + scale of the coarse
    octaves strictly and of the octave-0 rows as "found among ours", orientation on both parts of the file.  Returns the
    achieved fractions (they are also in every assertion message)."""
    gold = gold.astype(np.float64)
    mine = xys(pts)
    idx, dist = match_nearest(gold[:N_COARSE, :3], mine, 1e-2)
    idx0, dist0 = match_nearest(gold[N_COARSE:, :3], mine, 1e-2)
    d = ang_diff(gold[:N_COARSE, 3], pts["orientation"][idx].astype(np.float64))
    found0 = dist0 < 1e-2
    d0 = ang_diff(gold[N_COARSE:, 3][found0], pts["orientation"][idx0[found0]].astype(np.float64))
    got = {
        "coarse_within_1e-2": float((dist < 1e-2).mean()), "coarse_within_1e-3": float((dist < 1e-3).mean()),
        "coarse_one_to_one": len(set(idx.tolist())) == N_COARSE,
        # (north_star's 1e-3 for the orientation, in degrees.  Until round 6 this was tracked, not gated: a third of the
        # rows were within it.  The cause was the texture model -- the unit's four bilinear WEIGHTS are 8-bit fixed point,
        # not only the two fractions (oracle_tex2d) -- and with that restated 99 % of the rows are; the rest are rows
        # whose own location differs from the golden one in the last bits and lands on another 1/256 fraction)
        "coarse_ori_lt_1e-3": float((d < 1e-3).mean()), "coarse_ori_lt_1e-2": float((d < 1e-2).mean()),
        "coarse_ori_lt_0.1": float((d < 0.1).mean()), "coarse_ori_lt_1": float((d < 1.0).mean()),
        "coarse_ori_median": float(np.median(d)),
        "oct0_rows": int(len(dist0)), "oct0_found_1e-2": float(found0.mean()),
        "oct0_found_1e-3": float((dist0 < 1e-3).mean()),
        "oct0_ori_lt_1e-3": float((d0 < 1e-3).mean()), "oct0_ori_lt_1e-2": float((d0 < 1e-2).mean()),
        "oct0_ori_lt_0.1": float((d0 < 0.1).mean()), "oct0_ori_lt_1": float((d0 < 1.0).mean()),
        "oct0_ori_median": float(np.median(d0)),
    }
    msg = "%s vs golden: %s" % (label, ", ".join("%s=%s" % (k, ("%.4f" % v) if isinstance(v, float) else v)
                                                  for k, v in got.items()))
    # octaves >= 1: every golden row found within 1e-2 (base pixels), >= 98 % within 1e-3, one to one, in coarse octaves
    assert got["coarse_within_1e-2"] == 1.0 and got["coarse_within_1e-3"] >= 0.98 and got["coarse_one_to_one"], msg
    assert (pts["subsampling"][idx] >= 2.0).all(), msg
    # octave 0: the golden subset (racy at the reference's 4096 cap) is contained in ours; its matches are octave-0 points
    assert got["oct0_found_1e-2"] >= 0.995, msg
    assert (pts["subsampling"][idx0[found0]] == 1.0).all(), msg
    # orientation, on BOTH parts of the file.  SURVEY.md section 8c recommended >= 90 % within 0.1 and >= 97 % within 1
    # degree as the ceiling of a distributional match; the fixed-point weight rule (round 6) makes it the north star's own
    # 1e-3: >= 99 % of the coarse rows and >= 98 % of the octave-0 rows within 1e-3 degree, medians below 1e-4
    assert got["coarse_ori_lt_1e-3"] >= 0.99 and got["coarse_ori_lt_0.1"] >= 0.995 and got["coarse_ori_lt_1"] >= 0.999, msg
    assert got["oct0_ori_lt_1e-3"] >= 0.98 and got["oct0_ori_lt_0.1"] >= 0.995 and got["oct0_ori_lt_1"] >= 0.999, msg
    assert got["coarse_ori_median"] < 1e-4 and got["oct0_ori_median"] < 1e-4, msg
    return got


def orientation_outliers(gold, pts, peaks, limit_deg=1.0):
    """Why do some golden rows sit more than `limit_deg` from our orientation?  For every matched golden row (coarse rows
    and the octave-0 rows found among ours) beyond the limit: the oracle's ratio of second to first smoothed-histogram
    peak (`peaks[:, 0]`) and whether the golden angle lies within one bin (11.25 deg) of the orientation the oracle's
    SECOND peak would have given (`peaks[:, 1]`).  The reference computes both peaks and keeps the larger
    (cuSIFT_D.cu:362-394): a row explained by the second peak is a near-tie that CUDA's arithmetic (texture-unit
    rounding, atomicAdd order, libm) decided the other way -- noise, not a modelling difference.  Non-gating."""
    gold = gold.astype(np.float64)
    mine = xys(pts)
    idx, dist = match_nearest(gold[:, :3], mine, 1e-2)
    ok = dist < 1e-2
    idx, g = idx[ok], gold[ok]
    d = ang_diff(g[:, 3], pts["orientation"][idx].astype(np.float64))
    out = np.flatnonzero(d > limit_deg)
    ratio = peaks[idx[out], 0].astype(np.float64)
    second = peaks[idx[out], 1].astype(np.float64)
    d2 = ang_diff(g[out, 3], second)
    at_second = np.isfinite(second) & (d2 < 11.25)
    near = d[out] < 11.25  # same peak, shifted by less than a bin: a sample moved between neighbouring bins
    rest = ~at_second & ~near
    all_ratio = peaks[idx, 0].astype(np.float64)
    return {
        "rows_matched": int(ok.sum()), "outliers_gt_%g_deg" % limit_deg: int(len(out)),
        "outliers_at_second_peak": int(at_second.sum()),
        "outliers_at_second_peak_ratio_gt_0.98": int((at_second & (ratio > 0.98)).sum()),
        "outliers_at_second_peak_ratio_gt_0.90": int((at_second & (ratio > 0.90)).sum()),
        "outliers_at_second_peak_min_ratio": float(ratio[at_second].min()) if at_second.any() else None,
        "outliers_within_a_bin_of_first_peak": int((near & ~at_second).sum()),
        "outliers_within_a_bin_median_deg": float(np.median(d[out][near & ~at_second])) if (near & ~at_second).any() else None,
        "outliers_unexplained": int(rest.sum()),
        "outliers_unexplained_rows": [(float(g[out][i, 0]), float(g[out][i, 1]), float(g[out][i, 3]),
                                       float(pts["orientation"][idx[out][i]]), float(second[i]), float(ratio[i]))
                                      for i in np.flatnonzero(rest)[:8]],
        "median_ratio_all_rows": float(np.nanmedian(all_ratio)),
        "median_ratio_outliers": float(np.nanmedian(ratio)) if len(out) else None,
    }
