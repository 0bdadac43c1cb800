#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the MI355X-native SIFT extraction path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]/[3]): a batch of 64 synthetic 1920x1080 8-bit-valued images per GPU
(seeded `tile` images pre-blurred to sigma 1.0), 5 octaves, initBlur=1.0, thresh=3.0, edge=10.
One "step" = one pass of the whole hot path over that batch: ScaleDown pyramid, 8 blurs + 7 DoG per
octave, extrema + refinement, orientation, 128-D descriptors -- SiftData left in HBM; with N>1 ranks the
step ends with the RCCL all-gatherv of SiftData so every rank holds all N*64 images' keypoints.
Inputs are resident in HBM before the timed region.  Weak scaling: 64 images per GPU at every N.
Consecutive steps alternate over --streams HIP streams (default 2, one extractor each), so that the HBM-bound
ScaleDown chain and the launch tails of one batch overlap the VALU-bound kernels of the next; every step is still
one complete pass over one batch, and the timed region is bracketed by device-wide synchronisation.

Prints ONE JSON line on rank 0 (see the driver contract in the task description), with two extra
objects: `roofline` (blur+DoG kernel: algorithmic bytes / HIP-event duration vs 8 TB/s) and
`cpu_baseline` (the CPU oracle timed on this box's host cores over a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# ROCr reads its flags once, at hsa_init -- i.e. at the first HIP call of the process -- so this has to be in the
# environment before torch (or libcusift_amd.so) touches the GPU: the host driver only supports dmabuf IPC, and
# without it RCCL's cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def octave_dims(w, h, n_oct):
    dims = [(w, h)]
    for _ in range(1, n_oct):
        w, h = w // 2, h // 2
        if w < 1 or h < 1:
            break
        dims.append((w, h))
    return dims


def algorithmic_bytes(w, h, n_oct, n_img):
    """per octave, blur+DoG 32 B/px, downsample 4 B/px in + 4 B/px out, extrema 28 B/px."""
    dims = octave_dims(w, h, n_oct)
    blur = sum(32 * a * b for a, b in dims) * n_img
    find = sum(28 * a * b for a, b in dims) * n_img
    down = sum(4 * dims[i][0] * dims[i][1] + 4 * dims[i + 1][0] * dims[i + 1][1] for i in range(len(dims) - 1)) * n_img
    return blur, down, find


def cpu_baseline(w, h, params_kw, preblur, budget_s):
    """The CPU oracle (a restatement of the cuSIFT algorithm -- NOT OpenCV, which this image lacks) timed on
    the host cores: one image per thread (the C code releases the GIL), bounded to ~budget_s of wall time."""
    import threading

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cusift_amd import synth
    from oracle_binding import Oracle  # checker / baseline only

    cores = os.cpu_count() or 1
    # the CPUs this process may actually use: the cgroup quota if there is one (a GPU box hands a 16-CPU share of its
    # 256 hardware threads to a job), else the affinity mask
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    # measured on the box (16-CPU quota): 8 / 16 / 24 / 32 / 64 threads -> 102 / 184 / 197 / 189 / 163 Mpix/s
    threads = int(os.environ.get("CUSIFT_CPU_THREADS", "0")) or max(1, min(cores, usable + usable // 2, 48))
    oracle = Oracle()
    imgs = [synth.tile(5000 + i, w, h, preblur) for i in range(threads)]
    oracle.extract(imgs[0], **params_kw)  # warm-up (page-in, first-touch)
    counts = [0] * threads
    done = [0] * threads
    t_end = time.perf_counter() + budget_s

    def work(i):
        while True:
            counts[i] += len(oracle.extract(imgs[i], **params_kw))
            done[i] += 1
            if time.perf_counter() >= t_end:
                break

    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    n_img = sum(done)
    return {
        "value": round(n_img * w * h / dt / 1e6, 3),
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d x %dx%d images (same generator/params), %d threads, %.1f s wall; CPU restatement of the "
                  "cuSIFT algorithm (oracle/sift_oracle.c), not OpenCV" % (n_img, w, h, threads, dt),
        "keypoints_per_s": round(sum(counts) / dt, 1),
        "host_cores": cores,
        "usable_cpus": usable,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--octaves", type=int, default=5)
    ap.add_argument("--init-blur", type=float, default=1.0)
    ap.add_argument("--thresh", type=float, default=3.0)
    ap.add_argument("--max-pts", type=int, default=32768)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-stage-timers", action="store_true", help="do not bracket stages with HIP events")
    ap.add_argument("--gather", choices=["p2p", "padded"], default="p2p")
    ap.add_argument("--two-stage", action="store_true",
                    help="time the reference's two-stage pipeline (DoG planes in HBM) instead of the fused detection")
    ap.add_argument("--no-two-stage", action="store_true", help="skip the roofline exhibit leg")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps alternate over (one extractor each): the HBM-bound ScaleDown chain and "
                         "the launch tails of one batch overlap the VALU-bound kernels of the next")
    ap.add_argument("--gather-depth", type=int, default=1,
                    help="N > 1: the all-gatherv of step i is issued after step i + depth has been enqueued")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the all-gatherv of SiftData even with one rank (exercises the RCCL path on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from cusift_amd import capi, synth
    from cusift_amd.batch import BatchExtractor
    from cusift_amd.dist import begin_allgather, finish_allgather

    # Rank 0 prints exactly ONE line on stdout.  Libraries write there too (RCCL prints a version banner on
    # communicator creation), so from here on file descriptor 1 points at stderr and the JSON line goes to a
    # duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    capi.lib()  # fail loudly if the HIP extension is missing -- there is no fallback
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    w, h, B = args.width, args.height, args.batch
    prm_kw = dict(num_octaves=args.octaves, init_blur=args.init_blur, peak_thresh=args.thresh, edge_thresh=10.0,
                  lowest_scale=0.0, subsampling=1.0, max_pts=args.max_pts, tex_frac_bits=8)
    # One extractor (context + arena + output slots) per stream; step i runs on stream i % E.  A step is still one
    # whole pass of the hot path over one batch -- consecutive steps merely overlap on the device.
    from cusift_amd.batch import PipelinedExtractor

    E = max(1, args.streams)
    # output slots per extractor: a slot is overwritten E * n_slots steps later, and the gather of step i (which
    # reads it) is issued -- and its packing waited for -- right after step i + gather_depth has been enqueued
    n_slots = max(2, args.gather_depth // E + 2) if use_dist else 1
    pipe = PipelinedExtractor(B, w, h, n_streams=E, n_slots=n_slots,
                              fused_detect=0 if args.two_stage else 1, **prm_kw)
    exs = pipe.extractors
    ex = exs[0]

    # ---- synthetic inputs, resident in HBM before anything is timed ----
    from concurrent.futures import ThreadPoolExecutor

    seeds = [1000 + rank * B + i for i in range(B)]
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        host = list(pool.map(lambda s: synth.tile(s, w, h, args.init_blur), seeds))
    d_imgs = ex.images_from_numpy(np.stack(host))
    del host

    # N > 1: the all-gatherv of step i runs on a side stream while step i+1 is extracted on the main stream
    # (two output slots); its one host read-back (the counts) then waits only for the side stream.
    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream() if use_dist else None
    packer = ex.make_packer(side_stream) if use_dist else None
    pending = []
    state = {"gathered": None}

    slot_free = {}  # (stream index, slot) -> event after which the slot's last gather no longer reads it

    def finish_one():
        ticket, key = pending.pop(0)
        with torch.cuda.stream(side_stream):
            state["gathered"] = finish_allgather(ticket, method=args.gather, packer=packer)
            done = torch.cuda.Event()
            done.record(side_stream)
        slot_free[key] = done

    def step():
        key = (pipe.submitted % E, (pipe.submitted // E) % pipe.n_slots)
        pts, cnt, ev = pipe.submit(d_imgs, ready=slot_free.pop(key, None))
        if use_dist:
            # phase 1 of the all-gatherv right away (counts exchange + async copy to pinned memory, no host wait);
            # phase 2 (pack + shard exchange) once `gather_depth` further steps have been enqueued, by which time
            # the counts have long arrived
            with torch.cuda.stream(side_stream):
                side_stream.wait_event(ev)
                pending.append((begin_allgather(pts, cnt, ex.max_pts, n_images_max=B), key))
            if len(pending) > args.gather_depth:
                finish_one()

    def drain():
        while pending:
            finish_one()
        if use_dist:
            main_stream.wait_stream(side_stream)
        for st in pipe.streams[1:]:
            main_stream.wait_stream(st)

    def fence():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_stage_timers:
        for x in exs:
            x.ctx.timing_enable(True)
            x.ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    gathered = state["gathered"]
    stage = None
    if not args.no_stage_timers:
        for x in exs:  # kernel spans of all streams (with E > 1 they overlap in time: their sum exceeds the wall time)
            t = x.ctx.timing_read()
            stage = t if stage is None else {k: (stage[k][0] + t[k][0], stage[k][1] + t[k][1]) for k in t}
            x.ctx.timing_enable(False)

    # ---- single-stream leg (not part of `value`): with E > 1 the kernel spans of the timed region overlap in time,
    # so the per-stage table and the pyramid rate come from the same steps run on ONE stream
    stage_overlapped = None
    single_stream_ms = None
    if E > 1 and not args.no_stage_timers:
        stage_overlapped = stage
        ex.ctx.timing_enable(True)
        ex.ctx.timing_reset()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ex.extract(d_imgs)
        torch.cuda.synchronize()
        single_stream_ms = (time.perf_counter() - t1) / args.steps * 1e3
        stage = ex.ctx.timing_read()
        ex.ctx.timing_enable(False)

    # ---- roofline exhibit leg (not part of `value`): the same steps through the reference's two-stage pipeline
    # (LaplaceMulti -> DoG planes in HBM -> FindPointsMulti), to time the blur+DoG kernel the north star names.
    stage2 = None
    if not args.no_stage_timers and not args.no_two_stage and ex.params.fused_detect:
        ex.params.fused_detect = 0
        for _ in range(2):
            ex.extract(d_imgs)
        fence()
        ex.ctx.timing_enable(True)
        ex.ctx.timing_reset()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            ex.extract(d_imgs)
        torch.cuda.synchronize()
        two_stage_elapsed = time.perf_counter() - t2
        stage2 = ex.ctx.timing_read()
        ex.ctx.timing_enable(False)
        ex.params.fused_detect = 1
        fence()

    # max over ranks
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    counts = ex.valid_counts()
    local_kp = int(counts.sum().item())
    kp = torch.tensor([local_kp], dtype=torch.int64, device=dev)
    if use_dist:
        dist.all_reduce(kp, op=dist.ReduceOp.SUM)
        total_gathered = int(gathered[2][-1])  # noqa
        assert total_gathered == int(kp.item()), (total_gathered, int(kp.item()))
    total_kp = int(kp.item())

    if rank == 0:
        K = args.steps
        ms_per_step = elapsed / K * 1e3
        total_pix = world * B * w * h
        out = {
            "metric": "Mpix/s pyramid + keypoints/s end-to-end, 1920x1080 batch",
            "value": round(total_pix / (elapsed / K) / 1e6, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "batch of %d x %dx%d images per GPU (BASELINE configs[2]; x%d GPUs = configs[3] shape), "
                            "%d octaves, initBlur=%.1f, thresh=%.1f, edge=10, maxPts=%d; full SIFT extraction "
                            "(pyramid+DoG, extrema, orientation, 128-D descriptor)%s"
                            % (B, w, h, world, args.octaves, args.init_blur, args.thresh, args.max_pts,
                               "; + all-gatherv of SiftData (%s)" % args.gather if use_dist else ""),
                "images_per_gpu": B,
                "parallelism": "image-sharded x%d" % world,
                "streams_per_gpu": E,
            },
            "keypoints_per_s": round(total_kp / (elapsed / K), 1),
            "keypoints_per_step": total_kp,
        }
        blur_b, down_b, find_b = algorithmic_bytes(w, h, args.octaves, B)
        out["config"]["pipeline"] = "two-stage (DoG in HBM)" if args.two_stage else "fused detection (DoG on chip)"

        def traffic_of(kernel):
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            try:
                return json.load(open(tpath))[kernel]["hbm_bytes_per_launch"]
            except Exception:
                return None

        def blur_roofline(st, note):
            lap_ms, lap_n = st["laplace_multi"]
            if lap_n == 0 or lap_ms <= 0:
                return None
            # per launch: mean algorithmic bytes / mean HIP-event duration over the launches (5 octaves x K steps)
            achieved = (blur_b * K / lap_n) / (lap_ms * 1e-3 / lap_n) / 1e9
            return {
                "kernel": "laplace_multi_fast_kernel (8 blurs + 7 DoG planes, 32 B/px algorithmic)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic_of("laplace_multi_fast_kernel"),
                "algorithmic_bytes_per_launch": int(blur_b * K / lap_n),
                "avg_launch_ms": round(lap_ms / lap_n, 5),
                "launches": lap_n,
                "note": note,
            }

        def stage_table(st):
            return {k: round(st[k][0] / K, 4) for k in ("scale_down", "detect_multi", "describe_all", "laplace_multi",
                                                         "find_points_multi", "compute_orientations",
                                                         "extract_descriptors", "total")}

        if stage is not None:
            out["stage_ms_per_step"] = stage_table(stage)
            if stage_overlapped is not None:
                out["single_stream_leg"] = {
                    "ms_per_step": round(single_stream_ms, 4),
                    "note": "the timed region alternates steps over %d streams; stage_ms_per_step, fused_detect and "
                            "pyramid_mpix_per_s are measured on the same steps run on one stream (HIP events per "
                            "launch), where kernel spans do not overlap" % E,
                    "timed_region_kernel_spans_ms_per_step": stage_table(stage_overlapped)}
            sd_ms = stage["scale_down"][0]
            if stage["detect_multi"][1] > 0:
                det_ms, det_n = stage["detect_multi"]
                model_b = blur_b + find_b  # what the two reference stages move: 32 + 28 B/px
                ach = (model_b * K / det_n) / (det_ms * 1e-3 / det_n) / 1e9
                out["fused_detect"] = {
                    "kernel": "detect_fused_kernel (LaplaceMulti+FindPointsMulti, DoG planes kept in registers)",
                    "avg_launch_ms": round(det_ms / det_n, 5), "launches": det_n,
                    "model_bytes_per_launch": int(model_b * K / det_n),
                    "model_GBps": round(ach, 1), "model_frac_of_hbm_peak": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic_of("detect_fused_kernel"),
                    "note": "VALU-bound: the 60 B/px of the two-stage model are not moved (source image only); "
                            "model_GBps can exceed the HBM peak by construction",
                }
                out["pyramid_mpix_per_s"] = round(B * w * h / ((sd_ms + det_ms) / K * 1e-3) / 1e6, 1)
                rl = blur_roofline(stage2, "measured in the two-stage leg of this run (same inputs, HIP events on the "
                                           "launching stream); the timed region itself uses the fused kernel") if stage2 else None
                if rl:
                    out["roofline"] = rl
                    out["two_stage_leg"] = {"ms_per_step": round(two_stage_elapsed / K * 1e3, 4),
                                            "stage_ms_per_step": stage_table(stage2),
                                            "find_points_GBps": round(find_b / (stage2["find_points_multi"][0] / K * 1e-3) / 1e9, 1)}
            else:
                rl = blur_roofline(stage, "measured over the timed region")
                if rl:
                    out["roofline"] = rl
                lap_ms, fp_ms = stage["laplace_multi"][0], stage["find_points_multi"][0]
                out["pyramid_mpix_per_s"] = round(B * w * h / ((sd_ms + lap_ms) / K * 1e-3) / 1e6, 1)
                out["find_points_GBps"] = round(find_b / (fp_ms / K * 1e-3) / 1e9, 1) if fp_ms > 0 else None
            out["scale_down_GBps"] = round(down_b / (sd_ms / K * 1e-3) / 1e9, 1) if sd_ms > 0 else None
        if world == 1 and args.cpu_seconds > 0:
            cpu_kw = dict(prm_kw)
            out["cpu_baseline"] = cpu_baseline(w, h, cpu_kw, args.init_blur, args.cpu_seconds)
    for x in exs:
        x.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
