"""MatchSiftData (the first caller after the path): fp32 MFMA bound."""
import time

import numpy as np

from .common import FP32_VALU_PEAK_TF


def match_leg(capi, ctx, n):
    """MatchSiftData (cusift_match) on n x n synthetic unit descriptors: 2*n*n*128 flop per call on the exact-fp32
    MFMA (v_mfma_f32_16x16x4_f32), priced against the fp32 matrix peak."""
    rng = np.random.default_rng(5)
    p = np.zeros(n, dtype=capi.SIFT_POINT_DTYPE)
    d = np.abs(rng.normal(size=(n, 128))).astype(np.float32)
    p["data"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    d1 = capi.DeviceBuffer.from_numpy(ctx, p)
    d2 = capi.DeviceBuffer.from_numpy(ctx, p[::-1].copy())
    for _ in range(3):
        ctx.match(d1.ptr, n, d2.ptr, n, 1)
    ctx.synchronize()
    reps = 10
    t = time.perf_counter()
    for _ in range(reps):
        ctx.match(d1.ptr, n, d2.ptr, n, 1)
    ctx.synchronize()
    dt = (time.perf_counter() - t) / reps
    got = d1.to_numpy(capi.SIFT_POINT_DTYPE, n)
    ok = bool((got["match"] == np.arange(n)[::-1]).all())  # every descriptor's best match is its own copy
    d1.free()
    d2.free()
    tf = 2.0 * n * n * 128 / dt / 1e12
    return {"workload": "%d x %d descriptors of 128 floats, L2 distance, best + second best per row" % (n, n),
            "ms_per_call": round(dt * 1e3, 4), "pairs_per_s": round(n * n / dt, 1),
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(tf / FP32_VALU_PEAK_TF, 4), "traffic": None,
                         "note": "fp32 matrix peak = fp32 vector peak on gfx950 (MI355X_MICROARCH.md)"},
            "self_match_ok": ok}
