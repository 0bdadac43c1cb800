"""`config_legs`: the BASELINE configs the timed region is not -- each a driver-run number in the one line.

  configs[0]  single 640x480 greyscale PGM, 3 octaves (the reference's own CPU-runnable case): the fixture through the
              reference's entry point shape -- host float image in, host SiftData out -- on the GPU; the CPU side of this
              config (the oracle on one core, keypoint counts compared) is in `cpu_baseline.configs0`, the only leg that
              may touch oracle/
  configs[1]  single 1920x1080 image, 5 octaves, initBlur = 1.0, thresh = 3.0: latency of one extraction (eager launches
              and the recorded hipGraph), the back-to-back rate without a host wait per frame, and host float in -> host
              SiftData out (SiftData::Extract's shape, test/detector.cpp:37-49)
  configs[4]  single 8192x8192: whole on one GPU (rank 0), and -- when N > 1 ranks are up -- strip-tiled over them with a
              halo exchange per octave and an all-gatherv of the merged SiftData (cusift_tiled_* over the live
              communicator), the merged keypoint count asserted equal to the one-GPU run's; `tiled_model` is the committed
              prediction it is to be read against.
"""
import time

import numpy as np

from .models import tiled_model

TILED_W = TILED_H = 8192
TILED_SEED = 4242
TILED_MAX_PTS = 1 << 19


def single_frame(capi, synth, device, w=1920, h=1080, iters=200, warmup=20, host_iters=50):
    """BASELINE configs[1].  No torch: the C ABI only (tools/bench_latency.py prints the same dictionary)."""
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=32768)
    img = synth.tile(1000, w, h, preblur=1.0)
    p = capi.ialign_up(w, 128)
    src = np.zeros((h, p), dtype=np.float32)
    src[:, :w] = img
    out = {"workload": "single %dx%d image, 5 octaves, initBlur=1.0, thresh=3.0 (BASELINE configs[1])" % (w, h)}
    with capi.Context(device) as ctx:  # owns a non-blocking stream
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
        # PCIe-inclusive: SiftData::Extract(float *host, w, h) -- dense host image in, SiftData back on the host
        # (cusift_extract_host), pageable as the reference's malloc'd buffers and pinned as include/cuSIFT.h's SiftData
        dense = np.ascontiguousarray(img, dtype=np.float32)
        for label, pinned in (("host_float_in_host_siftdata_out", False), ("host_float_in_pinned_host_siftdata_out", True)):
            if pinned:
                h_buf = capi.HostBuffer(prm.max_pts * capi.SIFT_POINT_BYTES) if hasattr(capi, "HostBuffer") else None
                if h_buf is None:
                    continue
                h_pts = h_buf.as_numpy(capi.SIFT_POINT_DTYPE, prm.max_pts)
            else:
                h_pts = np.zeros(prm.max_pts, dtype=capi.SIFT_POINT_DTYPE)
            for _ in range(5):
                n = ctx.extract_host(dense, prm, d_pts.ptr, h_pts)
            lat = []
            for _ in range(host_iters):
                t0 = time.perf_counter()
                n = ctx.extract_host(dense, prm, d_pts.ptr, h_pts)
                lat.append(time.perf_counter() - t0)
            out[label] = {"latency_ms_median": round(float(np.median(lat)) * 1e3, 4),
                          "latency_ms_p95": round(float(np.percentile(lat, 95)) * 1e3, 4), "keypoints": int(n),
                          "bytes_up": int(dense.nbytes), "bytes_down": int(n) * 588,
                          "Mpix_per_s": round(w * h / float(np.median(lat)) / 1e6, 1)}
            if pinned:
                del h_pts
                h_buf.free()
        lat = []
        for _ in range(host_iters):
            t0 = time.perf_counter()
            n = ctx.extract_host(dense, prm, d_pts.ptr, None)
            lat.append(time.perf_counter() - t0)
        out["host_float_in_device_siftdata"] = {"latency_ms_median": round(float(np.median(lat)) * 1e3, 4)}
        u8 = np.ascontiguousarray(img.astype(np.uint8))
        if hasattr(ctx, "image_u8_h2d"):
            lat = []
            for _ in range(host_iters):
                t0 = time.perf_counter()
                ctx.image_u8_h2d(d_img.ptr, p, u8)
                ctx.extract_batch(*args)
                ctx.synchronize()
                lat.append(time.perf_counter() - t0)
            out["host_u8_in_device_siftdata"] = {"latency_ms_median": round(float(np.median(lat)) * 1e3, 4)}
    return out


def fixture_640x480(capi, synth, device, iters=100):
    """BASELINE configs[0], GPU side: the 640x480 fixture, 3 octaves, the reference test's parameters
    (test/detector.cpp:42-48: initBlur = 0, thresh = 0.1, edge = 10), host float image in -> host SiftData out."""
    img = np.ascontiguousarray(synth.fixture_image(), dtype=np.float32)
    h, w = img.shape
    prm = capi.default_params(num_octaves=3, init_blur=0.0, peak_thresh=0.1, edge_thresh=10.0, max_pts=16384)
    with capi.Context(device) as ctx:
        d_pts = capi.DeviceBuffer(ctx, prm.max_pts * capi.SIFT_POINT_BYTES)
        h_pts = np.zeros(prm.max_pts, dtype=capi.SIFT_POINT_DTYPE)
        for _ in range(5):
            n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
        lat = []
        for _ in range(iters):
            t0 = time.perf_counter()
            n = ctx.extract_host(img, prm, d_pts.ptr, h_pts)
            lat.append(time.perf_counter() - t0)
        d_pts.free()
    med = float(np.median(lat))
    return {"workload": "single %dx%d greyscale PGM (tests/golden/gray1.pgm = the reference's test/data/gray1), 3 octaves, "
                        "initBlur=0, thresh=0.1, edge=10 (BASELINE configs[0]); host float image in, host SiftData out"
                        % (w, h),
            "keypoints": int(n), "hip_ms_per_image_median": round(med * 1e3, 4),
            "hip_ms_per_image_p95": round(float(np.percentile(lat, 95)) * 1e3, 4),
            "hip_Mpix_per_s": round(w * h / med / 1e6, 1),
            "cpu_side": "cpu_baseline.configs0: the CPU oracle on one core over the same image, keypoint counts compared"}


def whole_8192(R, steps=20, warm=20):
    """BASELINE configs[4] on ONE GPU: the image through the ordinary driver (n = 1), at steady clocks (`warm` untimed
    calls first: five calls from a cold start read 0.68-0.70 ms where sixty read 0.60, profiles/r06/ab_handover.txt)."""
    from cusift_amd.batch import BatchExtractor

    torch, capi, synth = R.torch, R.capi, R.synth
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=TILED_MAX_PTS)
    img = synth.tile(TILED_SEED, TILED_W, TILED_H, preblur=1.0)
    ex = BatchExtractor(1, TILED_W, TILED_H, params=prm)
    try:
        d_img = ex.images_from_numpy(img[None])
        for it in range(steps + warm):
            if it == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            ex.extract(d_img)
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / steps
        n_whole = int(ex.valid_counts().sum().item())
        # stage times of the same call (HIP events per launch)
        _, st = R.run_single_stream(ex, d_img, steps)
        stages = {k: round(st[k][0] / steps, 4) for k in ("scale_down", "detect_multi", "describe_all") if st[k][1]}
    finally:
        ex.close()
    return {"workload": "single %dx%d image, 5 octaves, initBlur=1.0, thresh=3.0 (BASELINE configs[4]), whole on one GPU"
                        % (TILED_W, TILED_H),
            "ms_per_image": round(whole * 1e3, 4), "Mpix_per_s": round(TILED_W * TILED_H / whole / 1e6, 1),
            "keypoints": n_whole, "keypoints_per_s": round(n_whole / whole, 1), "gpu_ms_by_stage": stages}


def tiled_8192_all_ranks(R, steps=5):
    """BASELINE configs[4] over the ranks that are up (every rank calls this; rank 0 gets the dictionary, the others
    None).  Uses the timed region's communicator -- the C ABI's, RCCL from libcusift_amd.so -- on its side stream."""
    torch, dist, capi, synth = R.torch, R.dist, R.capi, R.synth
    from cusift_amd.dist import SiftGatherer
    from cusift_amd.tiling import StripExtractor, run_distributed

    if R.comm is None or R.gatherer is None:
        return {"skipped": "the C-ABI communicator is not up (%s)" % R.gather_impl} if R.rank == 0 else None
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=TILED_MAX_PTS)
    img = synth.tile(TILED_SEED, TILED_W, TILED_H, preblur=1.0)  # every rank builds the image (0.8 s) and keeps its strip
    ext = StripExtractor(R.rank, R.world, TILED_W, TILED_H, prm, device=R.dev, comm=R.comm)
    try:
        b = ext.plan.bounds
        strip = torch.from_numpy(np.ascontiguousarray(img[b[R.rank]:b[R.rank + 1]])).to(R.dev)
        del img
        gat = SiftGatherer(R.comm, 1, ext.max_pts, region_cap=ext.max_pts, device=R.dev)
        totals = None
        multi = R.world > 1  # (one rank + --force-gather: a rehearsal of this leg, RCCL self send / recv, no process group)
        for it in range(steps + 4):
            if it == 4:
                if multi:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            pts, cnt = run_distributed(ext, strip)
            _, _, totals = gat.gather(pts, cnt)
        outside = ext.check()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / steps], dtype=torch.float64, device=R.dev)
        if multi:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        gat.close()
    finally:
        ext.close()
    if R.rank != 0:
        return None
    sec = float(dt.item())
    return {"workload": "single %dx%d image strip-tiled over %d GPUs: ScaleDown -> halo exchange per octave -> band "
                        "detection + description, all-gatherv of the merged SiftData (cusift_tiled_extract + "
                        "cusift_allgatherv over RCCL); every rank ends holding the merged SiftData"
                        % (TILED_W, TILED_H, R.world),
            "n_gpus": R.world, "ms_per_image": round(sec * 1e3, 4), "Mpix_per_s": round(TILED_W * TILED_H / sec / 1e6, 1),
            "keypoints_merged": int(sum(int(t) for t in totals)), "collapse_octave": ext.plan.collapse,
            "keypoints_whose_footprint_left_the_halo": int(outside)}


def virtual_ranks_8192(R, P=8, steps=3):
    """One GPU: P virtual ranks run one after the other (the exchanges become device copies) -- what ONE rank's share of
    the tiled image costs when the exchange is free.  Includes each rank's D2H of its results."""
    torch, capi, synth = R.torch, R.capi, R.synth
    from cusift_amd.tiling import StripExtractor, run_virtual

    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=TILED_MAX_PTS)
    img = synth.tile(TILED_SEED, TILED_W, TILED_H, preblur=1.0)
    full = torch.from_numpy(img).to(R.dev)
    rows = TILED_H // P
    exts = [StripExtractor(k, P, TILED_W, TILED_H, prm, device=R.dev) for k in range(P)]
    try:
        strips = [full[k * rows:(k + 1) * rows] for k in range(P)]
        for it in range(steps + 1):
            if it == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            parts = run_virtual(exts, strips)
        torch.cuda.synchronize()
        virt = (time.perf_counter() - t0) / steps
    finally:
        for e in exts:
            e.close()
    return {"virtual_ranks": P, "total_ms": round(virt * 1e3, 4), "per_rank_ms_estimate": round(virt * 1e3 / P, 4),
            "keypoints_tiled": int(sum(len(p) for p in parts)),
            "note": "virtual ranks share one GPU, run one after the other and include the D2H of their results; the "
                    "estimate is total / P"}


def run_rank0(R):
    """The rank-0 part: configs[0], configs[1], the whole 8192^2 image, the virtual ranks, `tiled_model`.  With N > 1 the
    tiled leg runs AFTER this one (every rank takes part) and merge_tiled() adds its result."""
    tiled = None
    out, capi, synth = R.out, R.capi, R.synth
    K = R.args.steps
    legs = {}
    with R.leg_guard("configs[0]"):
        legs["configs[0]"] = fixture_640x480(capi, synth, R.local_rank, iters=max(20, 2 * K))
    with R.leg_guard("configs[1]"):
        legs["configs[1]"] = single_frame(capi, synth, R.local_rank, iters=max(40, 4 * K), warmup=10,
                                          host_iters=max(20, K))
    whole = None
    with R.leg_guard("configs[4]"):
        whole = whole_8192(R, steps=max(10, K))
        c4 = {"whole_on_one_gpu": whole}
        if not R.args.profile_run:
            c4["virtual_ranks_on_one_gpu"] = virtual_ranks_8192(R)
            c4["virtual_ranks_on_one_gpu"]["keypoints_equal_whole_image"] = (
                c4["virtual_ranks_on_one_gpu"]["keypoints_tiled"] == whole["keypoints"])
        if tiled is not None:
            c4["tiled_over_the_ranks"] = tiled
            if "keypoints_merged" in tiled:
                tiled["keypoints_equal_whole_image"] = tiled["keypoints_merged"] == whole["keypoints"]
                tiled["speedup_over_one_gpu"] = round(whole["ms_per_image"] / tiled["ms_per_image"], 3)
                if not tiled["keypoints_equal_whole_image"]:
                    out.setdefault("leg_errors", {})["configs[4] tiled"] = (
                        "merged keypoint count %d != whole-image count %d" % (tiled["keypoints_merged"], whole["keypoints"]))
        legs["configs[4]"] = c4
    if whole is not None:
        out["tiled_model"] = tiled_model(TILED_W, TILED_H, 5, whole["ms_per_image"], whole["keypoints"])
        if tiled is not None and "ms_per_image" in tiled:
            row = out["tiled_model"]["ranks"].get(str(R.world))
            if row:
                out["tiled_model"]["measured_ms_per_image_at_%d_ranks" % R.world] = tiled["ms_per_image"]
                out["tiled_model"]["measured_minus_predicted_ms_best_corner"] = round(
                    tiled["ms_per_image"] - row["latency_20us_eff_1.0"]["predicted_ms_per_image"], 4)
    out["config_legs"] = legs
    # the three as top-level numbers too
    if "configs[1]" in legs:
        out["single_frame_latency_ms"] = legs["configs[1]"]["eager"]["latency_ms_median"]
        out["single_frame_back_to_back_mpix_per_s"] = legs["configs[1]"]["eager"]["back_to_back_mpix_per_s"]
    if whole is not None:
        out["single_8192_ms"] = whole["ms_per_image"]
        out["single_8192_mpix_per_s"] = whole["Mpix_per_s"]


def merge_tiled(R, tiled):
    """Rank 0, after tiled_8192_all_ranks: the measured N-rank figure beside the whole-image one and the prediction."""
    out = R.out
    if tiled is None:
        return
    c4 = out.setdefault("config_legs", {}).setdefault("configs[4]", {})
    c4["tiled_over_the_ranks"] = tiled
    whole = c4.get("whole_on_one_gpu")
    if whole and "keypoints_merged" in tiled:
        tiled["keypoints_equal_whole_image"] = tiled["keypoints_merged"] == whole["keypoints"]
        tiled["speedup_over_one_gpu"] = round(whole["ms_per_image"] / tiled["ms_per_image"], 3)
        if not tiled["keypoints_equal_whole_image"]:
            out.setdefault("leg_errors", {})["configs[4] tiled"] = (
                "merged keypoint count %d != whole-image count %d" % (tiled["keypoints_merged"], whole["keypoints"]))
    row = out.get("tiled_model", {}).get("ranks", {}).get(str(R.world))
    if row and "ms_per_image" in tiled:
        out["tiled_model"]["measured_ms_per_image_at_%d_ranks" % R.world] = tiled["ms_per_image"]
        out["tiled_model"]["measured_minus_predicted_ms_best_corner"] = round(
            tiled["ms_per_image"] - row["latency_20us_eff_1.0"]["predicted_ms_per_image"], 4)
