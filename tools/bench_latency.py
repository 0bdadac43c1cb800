#!/usr/bin/env python3
"""BASELINE configs[1]: a single 1920x1080 image, 5 octaves, initBlur=1.0, thresh=3.0 on one MI355X.

Latency of one extraction (device-resident image -> SiftData in HBM, host notified), eager launches versus the
recorded hipGraph (cusift_graph_*), and the back-to-back rate without a host wait per frame.  No torch: the C ABI
only.  Prints one JSON line (informational; the driver's headline bench is bench.py).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(w=1920, h=1080, iters=200, warmup=20):
    from cusift_amd import capi, synth

    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=32768)
    img = synth.tile(1000, w, h, preblur=1.0)
    p = capi.ialign_up(w, 128)
    src = np.zeros((h, p), dtype=np.float32)
    src[:, :w] = img
    out = {"workload": "single %dx%d image, 5 octaves, initBlur=1.0, thresh=3.0 (BASELINE configs[1])" % (w, h)}
    with capi.Context(0) as ctx:  # owns a non-blocking stream
        d_img = capi.DeviceBuffer.from_numpy(ctx, src)
        d_pts = capi.DeviceBuffer(ctx, prm.max_pts * capi.SIFT_POINT_BYTES)
        d_cnt = capi.DeviceBuffer(ctx, 4)
        args = (d_img.ptr, 1, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
        ctx.extract_batch(*args)
        ctx.synchronize()
        out["keypoints"] = int(d_cnt.to_numpy(np.uint32, (1,))[0])
        graph = ctx.record_graph(*args)
        out["graph_nodes"] = graph.nodes

        def timed(fn, sync_each):
            for _ in range(warmup):
                fn()
            ctx.synchronize()
            lat = []
            t_all = time.perf_counter()
            for _ in range(iters):
                t0 = time.perf_counter()
                fn()
                if sync_each:
                    ctx.synchronize()
                    lat.append(time.perf_counter() - t0)
            ctx.synchronize()
            total = time.perf_counter() - t_all
            return lat, total

        for name, fn in (("eager", lambda: ctx.extract_batch(*args)), ("graph", graph.launch)):
            lat, _ = timed(fn, True)
            lat = np.array(lat) * 1e3
            _, total = timed(fn, False)
            out[name] = {"latency_ms_median": round(float(np.median(lat)), 4),
                         "latency_ms_p95": round(float(np.percentile(lat, 95)), 4),
                         "back_to_back_ms_per_frame": round(total / iters * 1e3, 4),
                         "back_to_back_mpix_per_s": round(w * h * iters / total / 1e6, 1)}
        # GPU time of one frame (stage timers: HIP events on the stream, eager path)
        ctx.timing_enable(True)
        ctx.timing_reset()
        for _ in range(20):
            ctx.extract_batch(*args)
        t = ctx.timing_read()
        ctx.timing_enable(False)
        out["gpu_ms_per_frame_by_stage"] = {k: round(v[0] / 20, 4) for k, v in t.items() if v[1]}
        graph.close()
        # PCIe-inclusive: SiftData::Extract(float *host, w, h) -- dense pageable host image in, SiftData back on the
        # host (cusift_extract_host), and the 8-bit upload front-end (1 byte per pixel + conversion on the device)
        h_pts = np.zeros(prm.max_pts, dtype=capi.SIFT_POINT_DTYPE)
        dense = np.ascontiguousarray(img, dtype=np.float32)
        for _ in range(5):
            n = ctx.extract_host(dense, prm, d_pts.ptr, h_pts)
        lat = []
        for _ in range(50):
            t0 = time.perf_counter()
            n = ctx.extract_host(dense, prm, d_pts.ptr, h_pts)
            lat.append(time.perf_counter() - t0)
        out["host_float_in_host_siftdata_out"] = {"latency_ms_median": round(float(np.median(lat)) * 1e3, 4),
                                                   "keypoints": int(n),
                                                   "bytes_up": int(dense.nbytes), "bytes_down": int(n) * 588}
        lat = []
        for _ in range(50):
            t0 = time.perf_counter()
            n = ctx.extract_host(dense, prm, d_pts.ptr, None)
            lat.append(time.perf_counter() - t0)
        out["host_float_in_device_siftdata"] = {"latency_ms_median": round(float(np.median(lat)) * 1e3, 4)}
        u8 = np.ascontiguousarray(img.astype(np.uint8))
        if hasattr(ctx, "image_u8_h2d"):
            lat = []
            for _ in range(50):
                t0 = time.perf_counter()
                ctx.image_u8_h2d(d_img.ptr, p, u8)
                ctx.extract_batch(*args)
                ctx.synchronize()
                lat.append(time.perf_counter() - t0)
            out["host_u8_in_device_siftdata"] = {"latency_ms_median": round(float(np.median(lat)) * 1e3, 4)}
    return out


if __name__ == "__main__":
    print(json.dumps(measure(iters=int(os.environ.get("LAT_ITERS", "200")))))
