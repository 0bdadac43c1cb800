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
