"""SiftData made host-visible (records packed on the device, copied to pinned memory on a copy stream, overlapped with
the next steps), host-to-host (the batch starts as pixels in pinned host memory and is uploaded every step), and the same
pipeline through the C ABI alone (cusift_pipe_*)."""
import time

import numpy as np


def run_host(R):
    args, torch, capi, out, ex, pipe, d_imgs, K = R.args, R.torch, R.capi, R.out, R.ex, R.pipe, R.d_imgs, R.args.steps
    B, E, dev, local_rank, local_kp = R.B, R.E, R.dev, R.local_rank, R.local_kp
    ex.params.concurrent_batches = E
    out["host_visible_leg"] = host_visible_leg(torch, capi, pipe, d_imgs, K, B, args.max_pts, local_rank, dev,
                                               total_local_kp=local_kp)
    out["keypoints_per_s_host_visible"] = out["host_visible_leg"]["keypoints_per_s"]
    # the same records without the 48 bytes extraction never writes (cusift_pack_points_trimmed: 540 bytes,
    # every field the reference fills, bit for bit -- expanded on the host by cusift_expand_trimmed_host)
    out["host_visible_trimmed_leg"] = host_visible_leg(torch, capi, pipe, d_imgs, K, B, args.max_pts, local_rank,
                                                       dev, total_local_kp=local_kp, compact="trimmed")
    out["keypoints_per_s_host_visible_trimmed"] = out["host_visible_trimmed_leg"]["keypoints_per_s"]
    # the optional 160-byte wire record (exact header fields, 8-bit descriptor with one step per record):
    # D2H no longer bounds the step; the exact 588-byte path above stays the default
    out["host_visible_compact_leg"] = host_visible_leg(torch, capi, pipe, d_imgs, K, B, args.max_pts, local_rank,
                                                       dev, total_local_kp=local_kp, compact=True)
    out["keypoints_per_s_host_visible_compact"] = out["host_visible_compact_leg"]["keypoints_per_s"]
    ex.params.concurrent_batches = 1


def run_host_in(R):
    args, torch, capi, out, ex, pipe, d_imgs, K = R.args, R.torch, R.capi, R.out, R.ex, R.pipe, R.d_imgs, R.args.steps
    B, w, h, E, dev, local_rank, local_kp = R.B, R.w, R.h, R.E, R.dev, R.local_rank, R.local_kp
    np_imgs, prm_kw, ms_per_step = R.np_imgs, R.prm_kw, R.ms_per_step
    ex.params.concurrent_batches = E
    h2h = {}
    u8_np = np.clip(np.rint(np_imgs), 0, 255).astype(np.uint8)
    variants = (("u8", torch.from_numpy(u8_np).pin_memory()), ("f32", torch.from_numpy(np_imgs).pin_memory()))
    for tag, h_src in variants:
        leg = host_visible_leg(torch, capi, pipe, d_imgs, K if tag == "u8" else max(8, K // 4), B, args.max_pts,
                               local_rank, dev, total_local_kp=local_kp, h_src=h_src)
        # the three things that can bound a step: the upload, the extraction, the read-back -- each as measured
        # in this run (PCIe rates with the leg's own buffers, each direction ALONE -- in the leg the two directions
        # run at once and share the host side of the link, so this bound is optimistic; extraction = the timed
        # region)
        parts = {"h2d_ms": leg["h2d_bytes_per_step"] / (leg["h2d_alone_GBps"] * 1e9) * 1e3,
                 "extract_ms": ms_per_step,
                 "d2h_ms": leg["d2h_bytes_per_step"] / (leg["d2h_alone_GBps"] * 1e9) * 1e3}
        bound = max(parts.values())
        leg["bound"] = {k: round(v, 4) for k, v in parts.items()}
        leg["bound"]["slowest"] = max(parts, key=parts.get)
        leg["bound"]["frac_of_bound"] = round(bound / leg["ms_per_step"], 4)
        h2h[tag] = leg
        del h_src
    # the same pipeline through the C ABI alone (cusift_pipe_*: what a C / C++ caller links against -- no
    # torch stream, event or tensor anywhere in it); the frames are the pinned u8 batch above
    try:
        torch.cuda.synchronize()
        depth_c = 4
        cpipe = capi.Pipe(local_rank, B, w, h, capi.default_params(**prm_kw), capi.PIPE_U8, depth=depth_c,
                          records_capacity=int(max(1.5 * local_kp, 4096)))
        frames_c = variants[0][1].numpy()  # a view of the pinned tensor
        n_c = max(8, K // 2)

        def run_c(steps):
            got = 0
            for _ in range(steps):
                if cpipe.in_flight() == depth_c:
                    got += len(cpipe.collect()[0])
                cpipe.submit(frames_c)
            while cpipe.in_flight():
                got += len(cpipe.collect()[0])
            return got

        run_c(depth_c)
        t1 = time.perf_counter()
        got_c = run_c(n_c)
        dt_c = time.perf_counter() - t1
        cpipe.close()
        h2h["u8_c_abi"] = {
            "ms_per_step": round(dt_c / n_c * 1e3, 4), "Mpix_per_s": round(B * w * h / (dt_c / n_c) / 1e6, 1),
            "keypoints_per_s": round(got_c / dt_c, 1), "batches_in_flight": depth_c,
            "note": "cusift_pipe_create / _submit / _collect (cusift_amd/csrc/sift_pipe.hip): the same upload -> "
                    "8-bit to float -> extraction -> pack -> read-back pipeline inside the library, driven by one "
                    "host thread through the C ABI; pinned 8-bit frames in, SiftData in the pipeline's pinned "
                    "slots out"}
        out["end_to_end_host_u8_c_abi_mpix_per_s"] = h2h["u8_c_abi"]["Mpix_per_s"]
    except Exception as e:  # noqa: BLE001
        h2h["u8_c_abi"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if "u8" in h2h:
        same = bool(np.array_equal(u8_np.astype(np.float32), np_imgs))
        h2h["u8"]["images"] = ("the timed images as 8-bit pixels (what a decoded frame holds): " +
                               ("the generator rounds to integers, so they ARE the timed images" if same else
                                "rounded, so keypoints per step differ slightly from the timed region's"))
    out["host_to_host"] = h2h
    out["end_to_end_host_u8_mpix_per_s"] = h2h["u8"]["Mpix_per_s"]
    out["end_to_end_host_u8_keypoints_per_s"] = h2h["u8"]["keypoints_per_s"]
    out["end_to_end_host_f32_mpix_per_s"] = h2h["f32"]["Mpix_per_s"]
    del u8_np, variants
    ex.params.concurrent_batches = 1


def host_visible_leg(torch, capi, pipe, d_imgs, K, B, max_pts, device_index, dev, total_local_kp, compact=False,
                     h_src=None):
    """Steps as in the timed region, but each step's SiftData is packed on the device (pack stream) and copied to pinned
    host memory (copy stream) while the next steps are extracted; the region ends when the last record is on the host.
    The copy size is a host argument, so a step's counts travel first (4 bytes x images) and its records one step
    later, exactly sized -- no host wait on the extraction streams, and the copies run back to back on their own
    stream (they, not the GPU, bound this leg: ~99 MB per step over PCIe)."""
    # At most three extraction streams here: with the pack and the copy stream that is five busy streams on the command
    # processor's four compute pipes -- a sixth made the compact leg a lottery (92-131 M keypoints/s by run with 4 + 2
    # streams, depending on which queues shared a pipe; 125 M with 3 + 2).
    all_streams, all_extractors = pipe.streams, pipe.extractors
    # (round 6, profiles/r06/upload_link.md: four extraction streams here read 48.4 against 47.6 Gpix/s host to host --
    # within the run-to-run spread; three stay)
    pipe.streams, pipe.extractors = all_streams[:3], all_extractors[:3]
    for x in pipe.extractors:
        x.params.concurrent_batches = len(pipe.streams)
    try:
        return _host_visible_leg(torch, capi, pipe, d_imgs, K, B, max_pts, device_index, dev, total_local_kp, compact,
                                 h_src)
    finally:
        pipe.streams, pipe.extractors = all_streams, all_extractors
        for x in pipe.extractors:
            x.params.concurrent_batches = len(pipe.streams)
        torch.cuda.synchronize()


def _host_visible_leg(torch, capi, pipe, d_imgs, K, B, max_pts, device_index, dev, total_local_kp, compact, h_src=None):
    """h_src: None -- the input is the HBM-resident batch d_imgs (the `host` leg); a pinned host tensor [B, h, w], uint8
    or float32 -- every step UPLOADS its batch first (the `host_in` legs: what a caller of the reference's entry point,
    which takes a host image, cuSIFT.cu:61-62, gets).  8-bit pixels are converted on the device (cusift_u8_to_f32, the
    front-end of SURVEY section 8f rank 2) on the extraction stream of their step."""
    pack_stream, copy_stream = torch.cuda.Stream(), torch.cuda.Stream()
    ingest = h_src is not None
    n_in = 3  # input buffers in flight: upload of step i+1 and i+2 beside the extraction of step i
    if ingest:
        h2d_stream = torch.cuda.Stream()
        h_img, w_img, pitch = pipe.h, pipe.w, pipe.pitch
        as_u8 = h_src.dtype == torch.uint8
        d_in = [torch.zeros((B, h_img, pitch), dtype=torch.float32, device=dev) for _ in range(n_in)]
        d_u8 = [torch.empty((B, h_img, w_img), dtype=torch.uint8, device=dev) for _ in range(n_in)] if as_u8 else None
        ev_in_free = [None] * n_in  # the extraction that read input buffer b has finished
        in_bytes = h_src.numel() * h_src.element_size()

    def upload(i):
        """enqueue the upload of step i's batch; returns (device images, event after which they are complete)"""
        b = i % n_in
        with torch.cuda.stream(h2d_stream):
            if ev_in_free[b] is not None:
                h2d_stream.wait_event(ev_in_free[b])
            if as_u8:
                d_u8[b].copy_(h_src, non_blocking=True)
            elif pitch == w_img:
                d_in[b].copy_(h_src, non_blocking=True)  # dense rows == pitched rows: one copy
            else:
                d_in[b][:, :, :w_img].copy_(h_src, non_blocking=True)
            up = torch.cuda.Event()
            up.record(h2d_stream)
        return b, up

    cctx = capi.Context(device_index, stream=pack_stream.cuda_stream)
    cap = int(max(1.5 * total_local_kp, 4096))  # records per step the staging buffers hold
    # staging slots: a step's records leave depth - 2 steps after it was enqueued.  The exact records are bound by the
    # copy itself (99 MB per step over PCIe); the compact ones are not, and need the host to stay further ahead than the
    # 4-stream extraction pipeline is deep
    fmt = "compact" if compact is True else ("trimmed" if compact == "trimmed" else "exact")
    depth = 8 if fmt == "compact" else 4
    rec_bytes = capi.WIRE_FORMATS[fmt][1]
    pack = {"exact": cctx.pack_points, "trimmed": cctx.pack_points_trimmed, "compact": cctx.pack_points_compact}[fmt]
    rec_dtype = {"exact": capi.SIFT_POINT_DTYPE, "trimmed": capi.TRIMMED_POINT_DTYPE, "compact": capi.COMPACT_POINT_DTYPE}[fmt]
    packed = [torch.empty((cap, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(depth)]
    offs = [torch.zeros(B + 1, dtype=torch.int32, device=dev) for _ in range(depth)]
    h_offs = [torch.zeros(B + 1, dtype=torch.int32).pin_memory() for _ in range(depth)]
    h_rec = [torch.empty((cap, rec_bytes), dtype=torch.uint8).pin_memory() for _ in range(depth)]
    ev_counts = [torch.cuda.Event() for _ in range(depth)]
    ev_copied = [None] * depth  # staging buffer j may be packed into again after this
    ev_slot = {}
    inflight = []
    got = {"records": 0, "bytes": 0}
    E = len(pipe.streams)

    def complete(j):
        ev_counts[j].synchronize()  # fired long ago: further steps have been enqueued since
        total = int(h_offs[j][B])
        assert total <= cap, (total, cap)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev_counts[j])
            h_rec[j][:total].copy_(packed[j][:total], non_blocking=True)
            ev_copied[j] = torch.cuda.Event()
            ev_copied[j].record(copy_stream)
        got["records"] += total
        got["bytes"] += total * rec_bytes + 4 * (B + 1)

    def one(i):
        j = i % depth
        e = pipe.submitted % E
        key = (e, (pipe.submitted // E) % pipe.n_slots)
        imgs = d_imgs
        if ingest:
            b, up = upload(i)
            imgs = d_in[b]
            with torch.cuda.stream(pipe.streams[e]):
                pipe.streams[e].wait_event(up)
                if as_u8:
                    pipe.extractors[e].ctx.u8_to_f32(d_in[b].data_ptr(), pitch, d_u8[b].data_ptr(), w_img, h_img, w_img,
                                                     n_images=B)
        pts, cnt, ev = pipe.submit(imgs, ready=ev_slot.pop(key, None))
        if ingest:
            ev_in_free[b] = ev
        with torch.cuda.stream(pack_stream):
            pack_stream.wait_event(ev)
            if ev_copied[j] is not None:
                pack_stream.wait_event(ev_copied[j])
            pack(pts.data_ptr(), cnt.data_ptr(), B, max_pts, packed[j].data_ptr(), cap, offs[j].data_ptr())
            done = torch.cuda.Event()
            done.record(pack_stream)  # the slot's records have been packed: the slot may be overwritten
            h_offs[j].copy_(offs[j], non_blocking=True)
            ev_counts[j].record(pack_stream)
        ev_slot[key] = done
        inflight.append(j)
        if len(inflight) > depth - 2:
            complete(inflight.pop(0))

    for i in range(depth):
        one(i)
    while inflight:
        complete(inflight.pop(0))
    torch.cuda.synchronize()
    got["records"] = got["bytes"] = 0
    t0 = time.perf_counter()
    for i in range(K):
        one(i)
    while inflight:
        complete(inflight.pop(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # spot check: the last step's host records are real (first record of image 0 has a finite, in-range location)
    rec = h_rec[(K - 1) % depth][:1].numpy().view(rec_dtype)
    assert np.isfinite(rec["coords2D"]).all() and rec["subsampling"][0] >= 1.0
    res = {"ms_per_step": round(dt / K * 1e3, 4), "keypoints_per_s": round(got["records"] / dt, 1),
           "d2h_GBps": round(got["bytes"] / dt / 1e9, 2), "d2h_bytes_per_step": int(got["bytes"] / K),
           "record_bytes": rec_bytes, "extraction_streams": E,
           "note": "device-resident input -> SiftData records in pinned host memory (packed on the device, copied on "
                   "a copy stream, overlapped with the following steps); bounded by the D2H copy when d2h_bytes_per_step "
                   "/ PCIe rate exceeds the extraction time"}
    if ingest:
        # each direction alone, same buffers and sizes: what PCIe gives this process on this box
        reps = max(4, K // 4)
        per_step = max(1, int(got["records"] / K))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for r in range(reps):
            with torch.cuda.stream(h2d_stream):
                (d_u8 if as_u8 else d_in)[r % n_in].copy_(h_src, non_blocking=True)
        torch.cuda.synchronize()
        h2d_alone = in_bytes * reps / (time.perf_counter() - t1)
        t1 = time.perf_counter()
        for r in range(reps):
            with torch.cuda.stream(copy_stream):
                h_rec[r % depth][:per_step].copy_(packed[r % depth][:per_step], non_blocking=True)
        torch.cuda.synchronize()
        d2h_alone = per_step * rec_bytes * reps / (time.perf_counter() - t1)
        px = B * pipe.w * pipe.h
        res.update({
            "input": "%d x %dx%d %s in pinned host memory, uploaded every step" % (B, pipe.w, pipe.h,
                                                                                   "uint8" if as_u8 else "float32"),
            "Mpix_per_s": round(px / (dt / K) / 1e6, 1),
            "keypoints_per_step": int(got["records"] / K),
            "h2d_bytes_per_step": int(in_bytes), "h2d_GBps": round(in_bytes * K / dt / 1e9, 2),
            "h2d_alone_GBps": round(h2d_alone / 1e9, 2), "d2h_alone_GBps": round(d2h_alone / 1e9, 2),
            "upload_buffers_in_flight": n_in,
            "note": "pinned host pixels -> H2D on an upload stream%s -> extraction (rotating over %d streams) -> records "
                    "packed on the device -> D2H on a copy stream into pinned host memory; upload, extraction and "
                    "read-back of consecutive steps overlap; the region ends when the last record is on the host"
                    % (" -> 8-bit to float on the device (cusift_u8_to_f32)" if as_u8 else "", E)})
    cctx.close()
    return res
