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
        # (north_star's 1e-3 for the orientation, in degrees: tracked, NOT gated -- it is met against the oracle, not
        # against the reference's own run: CUDA's arithmetic moves a third of the coarse rows by more than that)
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
    # orientation: distributional (hard histogram binning amplifies ulp noise; SURVEY.md hard part 1) -- on BOTH parts
    assert got["coarse_ori_lt_0.1"] >= 0.90 and got["coarse_ori_lt_1"] >= 0.97 and got["coarse_ori_median"] < 0.01, msg
    assert got["oct0_ori_lt_0.1"] >= 0.90 and got["oct0_ori_lt_1"] >= 0.97 and got["oct0_ori_median"] < 0.01, msg
    return got
