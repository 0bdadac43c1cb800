#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the MI355X-native SIFT extraction path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
(the first form with N > 1 starts the second one itself, as a child process, before anything touches a GPU)

Workload (BASELINE.json configs[2]/[3]): a batch of 64 synthetic 1920x1080 8-bit-valued images per GPU
(seeded `tile` images pre-blurred to sigma 1.0), 5 octaves, initBlur=1.0, thresh=3.0, edge=10.
One "step" = one pass of the whole hot path over that batch: ScaleDown pyramid (since round 5 written by the
detection launches themselves: CUSIFT_POLICY_PYRAMID_IN_DETECT, --pyramid-in-detect), 8 blurs + 7 DoG per
octave, extrema + refinement, orientation, 128-D descriptors -- SiftData left in HBM; with N>1 ranks the
step ends with the RCCL all-gatherv of SiftData (C ABI: cusift_allgatherv_*, ncclAllGather of the counts + one
ncclGroup of ncclSend/ncclRecv; 540-byte trimmed records expanded on arrival to 588-byte SiftPoint records unless
--gather-exact) so every rank holds all N*64 images' keypoints.
Before the W warm-up steps an untimed PRE-FLIGHT runs (--preflight, default 7 rounds = 28 steps, then blocks of 8 steps
until the step rate is steady: 52 steps in practice): every extractor runs the batch and all must report identical keypoint
counts; the rest keeps the device loaded so that the W + K steps do not start from idle clocks (tools/probe_rampup.py: after any idle gap the first ~20 ms of load run 5-12 % slow).
The line says what ran (config.preflight_steps) and carries the same K steps started from idle beside `value`
(ms_per_step_from_idle, value_from_idle_mpix_per_s); --preflight 0 measures without it.
Inputs are resident in HBM before the timed region.  Weak scaling: 64 images per GPU at every N.
Consecutive steps rotate over --streams HIP streams (default 4, one extractor each), so that the HBM-bound
ScaleDown chain and the launch tails of one batch overlap the VALU-bound kernels of the others (and the detection can
use tall row chunks: cusift_params.concurrent_batches); every step is still one complete pass over one batch, and the
timed region is bracketed by device-wide synchronisation.

Prints ONE JSON line on rank 0 (see the driver contract in the task description).  `value` comes from the timed
region only.  Everything else on the line is measured in separate, labelled legs after it (same inputs unless the
leg says otherwise), `--legs` selects them:
  single     the same steps on ONE stream with HIP events per launch: per-stage ms (kernel spans do not overlap)
  two_stage  the reference's LaplaceMulti -> DoG in HBM -> FindPointsMulti pipeline: `roofline` (blur+DoG kernel,
             algorithmic bytes / HIP-event time vs 8 TB/s, the north-star gate)
  host       SiftData made host-visible: packed records copied to pinned memory on a copy stream, overlapped with
             the next step (`keypoints_per_s_host_visible`; SURVEY.md section 8d's end-to-end definition)
  host_in    HOST to HOST: the batch starts as 8-bit pixels in pinned host memory (and, second variant, as float32 --
             what the reference's entry point takes, cuSIFT.cu:61-62), is uploaded every step, converted on the device,
             extracted, and its SiftData copied back to pinned memory; upload, extraction and read-back of consecutive
             steps overlap (`host_to_host`: ms/step, Mpix/s, keypoints/s, both PCIe rates, and the bound they set)
  repeat     the timed region four more times: min / median / max ms per step (the spread of `ms_per_step`)
  content    the same pipeline on other image content (`blobs`, un-pre-blurred `tile`), single stream AND pipelined
             like the timed region: keypoints/step, the fraction of octave-0 wave-rows the threshold pre-test skips,
             `value_blobs_mpix_per_s` / `value_tile_raw_mpix_per_s` -- how much of the rate is the images
  initblur0  the timed images with initBlur = 0.0 declared (the only value the reference's own test uses,
             test/detector.cpp:43): no identity levels in octave 0; `value_initblur0_mpix_per_s`
  ragged     64 x 1366x768 (no octave width is a multiple of 4): per-pixel rate next to 1080p's
  match      MatchSiftData (section 8 row f1) on 16384 x 16384 descriptors: TFLOP/s vs the fp32 MFMA peak
  cpu        `cpu_baseline`: the CPU oracle on this box's host cores over a bounded sample; OpenCV if importable
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
# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4).  The benchmark keeps four
# extraction streams busy and, with N > 1, a fifth for the exchange: on four queues the exchange stream shares a queue
# with an extraction stream, and its waits (for the producer's event, for the peers inside RCCL) block the kernels
# queued behind them -- measured at one rank (--force-gather --no-self-p2p): 1.325 ms per step on 4 queues, 1.233 on 8,
# against 1.157 without any exchange.  Read at HIP initialisation, so it has to be set here.  (With the exchange the
# bench also drops to three extraction streams: see `E` in main().)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
FP32_VALU_PEAK_TF = 157.3  # ibid. "Peak FP32 (vector)": 256 CUs x 4 SIMDs x 32 lanes x 2 flop (FMA) x 2.4 GHz
PRETEST_SKIP_HEADLINE = 0.73  # octave-0 wave-rows of the headline images the threshold pre-test skips (content leg)
ALL_LEGS = ("single", "repeat", "two_stage", "host", "host_in", "content", "initblur0", "ragged", "match", "cpu")


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


def usable_cpus():
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
    return cores, usable


def cpu_baseline(w, h, params_kw, preblur, budget_s):
    """The CPU oracle (a restatement of the cuSIFT algorithm -- NOT OpenCV) timed on the host cores: one image per
    thread (the C code releases the GIL), bounded to ~budget_s of wall time."""
    import threading

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cusift_amd import synth
    from oracle_binding import Oracle  # checker / baseline only

    cores, usable = usable_cpus()
    # measured on the box (16-CPU quota): 8 / 16 / 24 / 32 / 64 threads -> 102 / 184 / 197 / 189 / 163 Mpix/s
    threads = int(os.environ.get("CUSIFT_CPU_THREADS", "0")) or max(1, min(cores, usable + usable // 2, 48))
    oracle = Oracle()
    imgs = [synth.tile(5000 + i, w, h, preblur) for i in range(threads)]
    # wall-clock median of >= 5 runs after one warm-up.  A run = every thread extracts one image, all at the
    # same time (the C code releases the GIL); runs repeat until the budget is spent.
    counts = [0] * threads

    def one_pass():
        def work(i):
            counts[i] = len(oracle.extract(imgs[i], **params_kw))
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    one_pass()  # warm-up (page-in, first-touch)
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - t_start < budget_s and len(times) < 200):
        times.append(one_pass())
    dt = sum(times)
    n_img = threads * len(times)
    rates = sorted(threads * w * h / t / 1e6 for t in times)
    med = rates[len(rates) // 2]
    out = {
        "value": round(med, 3),
        "unit": "Mpix/s",
        "cores": threads,
        "kind": "port",
        "sample": "median of %d runs after 1 warm-up, each %d x %dx%d images at once on %d threads (same generator/"
                  "params), %.1f s wall in all; CPU restatement of the cuSIFT algorithm (oracle/sift_oracle.c), not OpenCV"
                  % (len(times), threads, w, h, threads, dt),
        "runs": len(times),
        "spread_mpix_per_s": {"min": round(rates[0], 3), "median": round(med, 3), "max": round(rates[-1], 3)},
        "keypoints_per_s": round(sum(counts) * med * 1e6 / (threads * w * h), 1),
        "host_cores": cores,
        "usable_cpus": usable,
    }
    out["opencv"] = opencv_baseline(imgs[: min(len(imgs), 8)], usable, min(budget_s, 8.0))
    return out


def opencv_baseline(imgs, threads, budget_s):
    """north_star asks for OpenCV's CPU SIFT beside the number (the reference's callers decode with OpenCV,
    test/detector.cpp:19-20).  It is not in this image; if a box has it, it is timed on the same images."""
    try:
        import cv2  # noqa: F401
    except Exception as e:  # ImportError, or a broken binary wheel
        return {"available": False, "note": "opencv: absent (import cv2: %s)" % type(e).__name__}
    try:
        cv2.setNumThreads(int(threads))
        sift = cv2.SIFT_create(0, 3, 0.04, 10, 1.6)
        u8 = [np.clip(i, 0, 255).astype(np.uint8) for i in imgs]
        sift.detectAndCompute(u8[0], None)
        n, kp, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            k, _ = sift.detectAndCompute(u8[n % len(u8)], None)
            kp += len(k)
            n += 1
        dt = time.perf_counter() - t0
        h, w = u8[0].shape
        return {"available": True, "version": cv2.__version__, "threads": int(threads),
                "Mpix_per_s": round(n * w * h / dt / 1e6, 3), "keypoints_per_s": round(kp / dt, 1),
                "note": "cv2.SIFT_create(0,3,0.04,10,1.6).detectAndCompute on the same images (8-bit)"}
    except Exception as e:
        return {"available": False, "note": "opencv: present but SIFT failed (%s)" % e}


def spawn_ranks(n, argv):
    """One rank per GPU as child processes (torch.distributed.run on 127.0.0.1, a free port); rank 0's JSON line is
    passed through on stdout, everything else the children print goes to stderr.  Returns the launcher's exit code
    (non-zero if any rank failed).  Never an exec: this process stays the parent."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no JSON line\n")
        rc = 1
    return rc


def dry_launch(args):
    """What the ranks do around the timed region, without a GPU: rendezvous (gloo), shard, barrier, max-over-ranks."""
    import torch
    import torch.distributed as dist

    from cusift_amd.dist import shard_range

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(world * args.batch, rank, world)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0 + 1e-6 * rank], dtype=torch.float64)
    n_img = torch.tensor([hi - lo], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_img, op=dist.ReduceOp.SUM)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "Mpix/s pyramid + keypoints/s end-to-end, 1920x1080 batch", "value": None,
                          "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_launch": True, "images_total": int(n_img.item()), "data": "none (launch rehearsal)"}),
              flush=True)
    return 0


XGMI_LINK_GBPS_PER_DIRECTION = 76.8  # one xGMI link of an MI355X: 153.6 GB/s bidirectional = 76.8 GB/s each way


def gather_model(kp_per_rank_step, rec_bytes, step_ms, lag_steps):
    """A PREDICTION of the all-gatherv step at 2 / 4 / 8 ranks, committed before any multi-GPU hardware has run it (no
    8-GPU node was reachable from the build box): the first measured scaling curve is to be read against these numbers.
    Exchange of one step: every rank sends its whole shard to each of its W - 1 peers, one dedicated point-to-point xGMI
    link per peer, all links at once (one ncclGroup of ncclSend / ncclRecv) -- so the time is one shard over one link,
    whatever W >= 2, and the same number of bytes arrives over the link's other direction."""
    shard = kp_per_rank_step * rec_bytes
    out = {"records_per_rank_per_step": int(kp_per_rank_step), "record_bytes": int(rec_bytes),
           "bytes_per_rank_per_step": int(shard), "bytes_per_peer_link_per_direction_per_step": int(shard),
           "assumed_link_GBps_per_direction": XGMI_LINK_GBPS_PER_DIRECTION,
           "assumed_rccl_p2p_efficiency": [1.0, 0.7],
           "extraction_ms_per_step": round(step_ms, 4), "finish_lags_begin_by_steps": lag_steps, "ranks": {}}
    for W in (2, 4, 8):
        row = {"bytes_received_per_rank_per_step": int(shard * (W - 1))}
        for eff in (1.0, 0.7):
            ex_ms = shard / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9 * eff) * 1e3
            row["eff_%.1f" % eff] = {
                "exchange_ms": round(ex_ms, 4),
                # own stream, finish lagging begin: latency is hidden, bandwidth is not -- a step cannot be shorter than
                # its exchange
                "ms_per_step_overlapped": round(max(step_ms, ex_ms), 4),
                "weak_scaling_efficiency_overlapped": round(step_ms / max(step_ms, ex_ms), 4),
                "ms_per_step_serial": round(step_ms + ex_ms, 4),
                "weak_scaling_efficiency_serial": round(step_ms / (step_ms + ex_ms), 4)}
        out["ranks"][str(W)] = row
    ex1 = shard / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3
    out["verdict"] = ("link-bound: one shard over one link takes %.2f ms at link peak against %.2f ms of extraction -- the "
                      "exchange, not the GPU, sets the step from 2 ranks up" % (ex1, step_ms)) if ex1 > step_ms else (
                      "extraction-bound at link peak (%.2f ms exchange against %.2f ms); link-bound below %.0f %% RCCL "
                      "efficiency" % (ex1, step_ms, 100.0 * ex1 / step_ms))
    out["not_modelled"] = ("the counts all-gather (a few tens of microseconds, hidden by the lag), the CUs RCCL's send / "
                           "receive kernels take from the extraction, HBM traffic of the arriving shards (%.2f GB per step "
                           "at 8 ranks: ~0.1 ms of HBM time)" % (shard * 7 / 1e9))
    out["options"] = {"compact 160-byte wire record (--gather-compact; 8-bit descriptor, lossy)":
                      round(kp_per_rank_step * 160 / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3, 4),
                      "trimmed 540-byte wire record (the N > 1 default since round 5: the 135 floats extraction writes, EXACT, "
                      "expanded on arrival to 588-byte SiftPoint records; --gather-exact keeps 588 on the wire: %.4f ms)"
                      % (kp_per_rank_step * 588 / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3):
                      round(kp_per_rank_step * 540 / (XGMI_LINK_GBPS_PER_DIRECTION * 1e9) * 1e3, 4),
                      "unit": "exchange ms per step at link peak"}
    return out


def load_profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a 60 ms timed region after 12 ms of warm-up -- region-to-region spread on one box is +-5 % at 20 steps
    # (clock management; `ms_per_step_spread`), and the legs below scale with K
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--octaves", type=int, default=5)
    ap.add_argument("--init-blur", type=float, default=1.0)
    ap.add_argument("--thresh", type=float, default=3.0)
    ap.add_argument("--max-pts", type=int, default=32768)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--legs", default=",".join(ALL_LEGS),
                    help="comma-separated extra legs after the timed region (%s); 'none' = only the timed region"
                         % ", ".join(ALL_LEGS))
    ap.add_argument("--two-stage", action="store_true",
                    help="time the reference's two-stage pipeline (DoG planes in HBM) instead of the fused detection")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams the steps alternate over (one extractor each): the HBM-bound ScaleDown chain and "
                         "the launch tails of one batch overlap the VALU-bound kernels of the next.  0 (default): four, "
                         "or three when the step includes the exchange (N > 1, --force-gather) -- with the exchange "
                         "stream that makes four busy streams, one per pipe of the command processor")
    ap.add_argument("--gather-capacity", type=int, default=8192,
                    help="N > 1: records per image the gathered SiftData buffer is sized for")
    ap.add_argument("--gather-compact", action="store_true",
                    help="N > 1: exchange 160-byte compact records (exact header fields, 8-bit descriptor) instead of the "
                         "exact 588-byte SiftPoint records -- NOT the default: the metric is the all-gatherv of SiftData")
    ap.add_argument("--gather-trimmed", action="store_true",
                    help="N > 1: exchange 540-byte trimmed records and LEAVE them trimmed (no expansion on arrival)")
    ap.add_argument("--gather-exact", action="store_true",
                    help="N > 1: exchange the 588-byte SiftPoint records as they are.  The default since round 5: the "
                         "records TRAVEL trimmed (540 bytes: the 135 floats extraction writes, exact; the other 12 are "
                         "never written by extraction and uninitialised in the reference, cuSIFT.cu:24,29) and are "
                         "EXPANDED ON ARRIVAL (cusift_expand_gathered), so every rank ends the step holding the 588-byte "
                         "SiftData of all images for 8 %% fewer bytes per xGMI link")
    ap.add_argument("--no-self-p2p", action="store_true",
                    help="with --force-gather at one rank: do not route the local shard through ncclSend/ncclRecv to self "
                         "(what remains is what every rank does for ITS OWN shard at any N: clamp, pack into its region, "
                         "counts all-gather, publish)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the all-gatherv of SiftData even with one rank (self send/recv: exercises the RCCL path "
                         "on one GPU)")
    ap.add_argument("--profile-run", action="store_true",
                    help="under rocprofv3 (tools/profile_gpu.sh): skip the sub-legs that launch a kernel outside the "
                         "5-octave cycle, so that per-launch PMC averages are over whole steps")
    ap.add_argument("--pyramid-in-detect", type=int, default=-1, choices=(-1, 0, 1, 2),
                    help="CUSIFT_POLICY_PYRAMID_IN_DETECT of every extraction context: -1 the library's default (calls of "
                         ">= 6 Mpixel: every detection writes the next octave's image, no ScaleDown launches), 0 the "
                         "ScaleDown chain first (the reference's order, cuSIFT.cu:175-192), 1 octave 0 only, 2 every octave")
    ap.add_argument("--preflight", type=int, default=7,
                    help="untimed set-up rounds before the W warm-up steps, each one batch per extractor output slot: the "
                         "first round is CHECKED (all extractors must report identical keypoint counts), the others only "
                         "keep the device loaded so that the W + K steps do not start from idle clocks (after any idle "
                         "gap the first ~20 ms of load run 5-12 %% slow: tools/probe_rampup.py).  0: no pre-flight at all "
                         "-- the timed region then measures the ramp (config.preflight_steps says what ran; the line "
                         "also carries ms_per_step_spread.ms_per_step_from_idle)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the launch only: ranks rendezvous over gloo on the CPU, shard the batch, barrier, "
                         "reduce a time and rank 0 prints a line -- no GPU, no extraction (tests the --gpus N spawn)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as a plain command: this process has not touched a GPU (no torch import, no HIP
        # call so far) and never will -- it starts the N ranks as CHILDREN and relays rank 0's line
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.dry_launch:
        raise SystemExit(dry_launch(args))
    legs = set() if args.legs in ("none", "") else set(x for x in args.legs.split(",") if x)
    unknown = legs - set(ALL_LEGS)
    if unknown:
        raise SystemExit("unknown legs: %s" % sorted(unknown))
    if args.cpu_seconds <= 0:
        legs.discard("cpu")

    import torch
    import torch.distributed as dist

    from cusift_amd import capi, synth
    from cusift_amd.batch import BatchExtractor, PipelinedExtractor
    from cusift_amd.dist import SiftGatherer, begin_allgather, finish_allgather, make_comm

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
    local_rank %= max(1, torch.cuda.device_count())  # rehearsals with more ranks than GPUs share devices (RCCL permitting)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_gather
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch.distributed carries the communicator's 128-byte id, the barrier and the max-over-ranks of the time;
        # the SiftData exchange itself is the C ABI's (RCCL called from libcusift_amd.so)
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    w, h, B = args.width, args.height, args.batch
    prm_kw = dict(num_octaves=args.octaves, init_blur=args.init_blur, peak_thresh=args.thresh, edge_thresh=10.0,
                  lowest_scale=0.0, subsampling=1.0, max_pts=args.max_pts, tex_frac_bits=8)
    # One extractor (context + arena + output slots) per stream; step i runs on stream i % E.  A step is still one
    # whole pass of the hot path over one batch -- consecutive steps merely overlap on the device.
    # The command processor has four compute pipes; hardware queues 1, 5, 9 ... share the first.  A fifth busy stream
    # (the exchange) lands on a pipe that an extraction stream uses and the two queues' packets wait for each other:
    # measured at one rank (--force-gather, self send/recv): 4 + 1 streams 1.32-1.33 ms per step, 3 + 1 streams 1.29
    # (without the exchange four streams win: 1.17 against 1.19).
    E = args.streams if args.streams > 0 else (3 if use_dist else 4)
    n_slots = 2 if use_dist else 1  # a slot is read by the pack of its step's gather while the next steps are extracted
    pipe = PipelinedExtractor(B, w, h, n_streams=E, n_slots=n_slots,
                              fused_detect=0 if args.two_stage else 1, **prm_kw)
    exs = pipe.extractors
    ex = exs[0]
    if args.pyramid_in_detect != -1:
        for x in exs:
            x.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, args.pyramid_in_detect)

    # ---- synthetic inputs, resident in HBM before anything is timed ----
    from concurrent.futures import ThreadPoolExecutor

    def make_images(fn, seeds):
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            return np.stack(list(pool.map(fn, seeds)))

    seeds = [1000 + rank * B + i for i in range(B)]
    np_imgs = make_images(lambda s: synth.tile(s, w, h, args.init_blur), seeds)
    d_imgs = ex.images_from_numpy(np_imgs)

    # N > 1: the all-gatherv of step i runs on a side stream (its own context + communicator).  begin(i) -- counts
    # exchange + the local shard packed into its region -- is enqueued right after step i; finish(i) -- the one host READ
    # of the counts, then the grouped ncclSend/ncclRecv -- after step i + LAG has been enqueued.  The host runs AHEAD of
    # the device (enqueueing a step takes ~0.1 ms, executing it ~1.2), so finish(i) usually finds the counts flag not yet
    # set and spins on it: that wait is the host's throttle, not device idle time -- the device still has LAG steps
    # queued.  config.gather_host_waits counts those finishes (round 3 measured 60 of 60) and config.gather_host_wait_ms
    # is the time spent in them.
    main_stream = torch.cuda.current_stream()
    side_stream = torch.cuda.Stream() if use_dist else None
    LAG = E
    gatherer = None
    gather_impl = None
    comm = side_ctx = None
    region_cap = B * args.gather_capacity
    if use_dist:
        # The exchange is the C ABI's (RCCL called from libcusift_amd.so).  On the build box it has met more than one
        # rank only over the in-process test transport (tests/test_multirank_gpu.py; RCCL refuses two ranks per GPU),
        # so a failure to bring the communicator up is not allowed to cost the run: all ranks then agree to fall back to
        # the torch.distributed twin of the same exchange, and the JSON line says which one ran (config.gather_impl).
        err = ""
        try:
            side_ctx = capi.Context(local_rank, stream=side_stream.cuda_stream)
            comm = make_comm(side_ctx, self_p2p=(world == 1 and not args.no_self_p2p))
            gatherer = SiftGatherer(comm, B, args.max_pts, region_cap=region_cap, device=dev, n_out=LAG + 2,
                                    depth=LAG + 1, wire_format="compact" if args.gather_compact else (
                                        "exact" if args.gather_exact else "trimmed"),
                                    expand=not (args.gather_compact or args.gather_exact or args.gather_trimmed))
        except Exception as e:  # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            gather_impl = ("C ABI (cusift_allgatherv_*): ncclAllGather of counts + one ncclGroup of ncclSend/ncclRecv, "
                           "finish lags begin by %d steps" % LAG)
        else:
            print("bench.py: C-ABI communicator unavailable (%s); using the torch.distributed exchange" % err,
                  file=sys.stderr)
            gatherer = None
            LAG = 1
            gather_impl = "torch.distributed fallback (C ABI communicator failed: %s)" % (err or "on another rank")
    pending = []
    state = {"gathered": None}
    slot_free = {}  # (stream index, slot) -> event after which the slot's records have been packed (it may be rewritten)
    packer = ex.make_packer(side_stream) if (use_dist and gatherer is None) else None

    def finish_one():
        key, ticket = pending.pop(0)
        with torch.cuda.stream(side_stream):
            if gatherer is not None:
                counts_h, buf, totals = gatherer.finish()
                state["gathered"] = (counts_h, buf, totals)
            else:
                ac, ga, off = finish_allgather(ticket, method="p2p", packer=packer)
                state["gathered"] = (ac, ga, np.diff(off.numpy()))
                done = torch.cuda.Event()
                done.record(side_stream)
                slot_free[key] = done

    def step():
        e = pipe.submitted % E
        key = (e, (pipe.submitted // E) % pipe.n_slots)
        pts, cnt, ev = pipe.submit(d_imgs, ready=slot_free.pop(key, None))
        if use_dist:
            ticket = None
            with torch.cuda.stream(side_stream):
                if gatherer is not None:
                    # ordered after the extraction by begin() itself; the slot is free again once its records sit in
                    # the gathered buffer (the event begin() returns)
                    slot_free[key] = gatherer.begin(pts, cnt, producer=exs[e].ctx)
                else:
                    side_stream.wait_event(ev)
                    ticket = begin_allgather(pts, cnt, ex.max_pts, n_images_max=B)
            pending.append((key, ticket))
            if len(pending) > LAG:
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
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region runs the PRODUCT: stage timers off (two hipEventRecord per launch, and a driver that pins the
    # per-octave launch sequence while they are on), whatever legs follow -- `--legs none` takes the same path, and
    # config.timed_region_forks / _timers say so.  The kernel-span table of the overlapped streams comes from a REPEAT of
    # the region with the timers on (`timed_region_kernel_spans_ms_per_step`, with that repeat's own ms per step).
    torch.cuda.synchronize()
    # Pre-flight, before the W warm-up steps: every extractor runs the batch a few times and all of them must report the
    # same keypoint counts (four contexts, four arenas, one answer) -- a set-up check, not part of the contract's W + K
    # steps and never timed.  It is also what takes the device out of its idle clocks: after ANY idle gap (50 ms is
    # enough) the first ~20 ms of load run 5-12 % slow (tools/probe_rampup.py: 1.09, 1.02, then 0.97 ms per step in
    # steady state; staggering the streams' starts changes nothing, a region that follows another without a gap starts
    # at the steady rate).  W = 5 steps is 5 ms.  config.preflight_steps says how many ran; the repeat leg reports the
    # same region started from idle (`ms_per_step_from_idle`) beside it.
    PREFLIGHT = max(0, args.preflight) * E * n_slots
    if PREFLIGHT:
        for _ in range(E * n_slots):
            pipe.submit(d_imgs)
        pipe.synchronize()
        ref_counts = exs[0].slots[0][1].clone()
        for x in exs:
            for _, cnt_t in x.slots:
                if not torch.equal(cnt_t, ref_counts):
                    raise SystemExit("bench.py: pre-flight: extractors disagree on the keypoint counts of the same batch")
    # (the comparison above is torch's first work in the process -- tens of milliseconds of lazy initialisation during
    # which the device idles -- so the load that takes it out of its idle clocks comes AFTER it, with nothing but
    # enqueueing between here and the timed region's fence)
    for _ in range(max(0, PREFLIGHT - E * n_slots)):
        pipe.submit(d_imgs)
    # ... and, because how long the ramp takes depends on how deep the device slept (one run in six of the driver's form read
    # 1.04 ms behind the fixed 28 steps where the others read 0.95), load continues in blocks of 2 E steps until two
    # consecutive blocks run within 2 % of each other and of the fastest block seen -- at most 24 more blocks.  The blocks'
    # ms per step are in the line (config.preflight_blocks_ms_per_step): what the device did before the W + K steps is on record.
    preflight_blocks = []
    if PREFLIGHT:
        pipe.synchronize()
        for _ in range(24):
            t_b = time.perf_counter()
            for _ in range(2 * E):
                pipe.submit(d_imgs)
            pipe.synchronize()
            preflight_blocks.append((time.perf_counter() - t_b) / (2 * E) * 1e3)
            PREFLIGHT += 2 * E
            if len(preflight_blocks) >= 3:
                a, b, best = preflight_blocks[-1], preflight_blocks[-2], min(preflight_blocks)
                if abs(a - b) <= 0.02 * best and max(a, b) <= 1.02 * best:
                    break
    for _ in range(args.warmup):
        step()
    fence()
    forks_before = sum(x.ctx.forks() for x in exs)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    forks_timed = sum(x.ctx.forks() for x in exs) - forks_before
    gathered = state["gathered"]
    gather_waits = (comm.host_waits(), comm.host_wait_ms()) if comm is not None else None
    stage_overlapped = None
    spans_ms_per_step = None
    if legs and not use_dist:
        for x in exs:
            x.ctx.timing_enable(True)
            x.ctx.timing_reset()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        spans_ms_per_step = (time.perf_counter() - t1) / args.steps * 1e3
        for x in exs:  # kernel spans of all streams (with E > 1 they overlap in time: their sum exceeds the wall time)
            t = x.ctx.timing_read()
            stage_overlapped = t if stage_overlapped is None else {
                k: (stage_overlapped[k][0] + t[k][0], stage_overlapped[k][1] + t[k][1]) for k in t}
            x.ctx.timing_enable(False)

    # N > 1: the same K steps WITHOUT the exchange, right behind the timed region -- what gather_model needs as the
    # extraction's own time, so that "measured - predicted" means something (max over ranks, like the timed region)
    extraction_only_ms = None
    if use_dist:
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pipe.submit(d_imgs)
        pipe.synchronize()
        eo = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(eo, op=dist.ReduceOp.MAX)
        extraction_only_ms = float(eo.item()) / args.steps * 1e3
    # max over ranks (and every rank's own time, for the line)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    per_rank_elapsed = [elapsed]
    if world > 1:
        every = [torch.zeros_like(el) for _ in range(world)]
        dist.all_gather(every, el)
        per_rank_elapsed = [float(t.item()) for t in every]
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    counts = ex.valid_counts()
    local_kp = int(counts.sum().item())
    kp = torch.tensor([local_kp], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(kp, op=dist.ReduceOp.SUM)
    total_kp = int(kp.item())
    if use_dist:
        total_gathered = int(np.asarray(gathered[2], dtype=np.int64).sum())
        assert total_gathered == total_kp, (total_gathered, total_kp)

    K = args.steps
    out = None
    if rank == 0:
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
                            "(pyramid+DoG, extrema, orientation, 128-D descriptor), SiftData left in HBM%s"
                            % (B, w, h, world, args.octaves, args.init_blur, args.thresh, args.max_pts,
                               "; + all-gatherv of SiftData every step (C ABI over RCCL: counts all-gather + grouped "
                               "send/recv)"
                               if use_dist else ""),
                "images_per_gpu": B,
                "parallelism": "image-sharded x%d" % world,
                "streams_per_gpu": E,
                "pipeline": "two-stage (DoG in HBM)" if args.two_stage else "fused detection (DoG on chip)",
                "timed_region_timers": False,
                "timed_region_forks": int(forks_timed),
                "preflight_steps": PREFLIGHT,
                "preflight_blocks_ms_per_step": [round(x, 4) for x in preflight_blocks],
                "preflight_note": "untimed set-up check before the W warm-up steps: every extractor runs the batch, all "
                                  "must report identical keypoint counts; it also takes the device out of its idle clocks "
                                  "(ms_per_step_spread.ms_per_step_from_idle = the same region started 50 ms after idle)",
                "pyramid_in_detect": ex.ctx.get_policy(capi.POLICY_PYRAMID_IN_DETECT),
                "pyramid_in_detect_note": "-1 = the library's default: a call of >= 6 Mpixel searches its octaves finest "
                                          "first and every detection launch also writes the next octave's image "
                                          "(ScaleDown's arithmetic, bit for bit) -- no ScaleDown launch, no memset",
            },
            "keypoints_per_s_in_hbm": round(total_kp / (elapsed / K), 1),
            "keypoints_per_step": total_kp,
        }
        # (N = 1: the prediction is for the wire format an N > 1 run of this command line would use)
        rec_b = gatherer.record_bytes if gatherer is not None else (
            160 if args.gather_compact else (588 if args.gather_exact else 540))
        out["gather_model"] = gather_model(local_kp, rec_b, extraction_only_ms if use_dist else ms_per_step,
                                           LAG if use_dist else E)
        if use_dist:
            out["gather_model"]["extraction_ms_per_step_source"] = (
                "the same K steps run without the exchange right behind the timed region (max over ranks)")
        out["gather_model"]["step_of_this_run_includes_an_exchange"] = bool(use_dist)
        if use_dist:
            out["config"]["gather_impl"] = gather_impl
            out["config"]["rccl_library"] = capi.Comm.library()
            out["config"]["gather_region_records"] = region_cap
            out["config"]["gather_record_bytes"] = rec_b
            out["config"]["gather_wire_format"] = (
                "%s%s" % (gatherer.wire_format, ", expanded on arrival to 588-byte SiftPoint records" if gatherer.expand
                          else "")) if gatherer is not None else "exact (torch.distributed fallback)"
            # what the LIBRARY reports (ncclCommCount / ncclGetVersion), not this script's own bookkeeping: "RCCL saw N
            # ranks" can be read off the line
            info = comm.info() if comm is not None else {}
            out["config"]["rccl_ranks"] = info.get("lib_ranks")
            out["config"]["rccl_version"] = info.get("lib_version")
            out["config"]["ms_per_step_by_rank"] = [round(float(t) / K * 1e3, 4) for t in per_rank_elapsed]
            ex_ms = out["gather_model"]["ranks"].get(str(world), {}).get("eff_1.0", {}).get("ms_per_step_overlapped")
            if ex_ms:
                out["gather_model"]["measured_minus_predicted_ms_at_link_peak"] = round(ms_per_step - ex_ms, 4)
            if gather_waits is not None:
                out["config"]["gather_host_waits"] = gather_waits[0]
                out["config"]["gather_host_wait_ms"] = round(gather_waits[1], 3)

    # ================================================================================================================
    # Extra legs (rank 0's GPU only; not part of `value`).  With N > 1 the other ranks wait at the final barrier.
    # ================================================================================================================
    blur_b, down_b, find_b = algorithmic_bytes(w, h, args.octaves, B)
    traffic = load_profile_json("traffic.json")
    valu = load_profile_json("valu.json")
    isa_mix = load_profile_json("isa_mix.json")

    def stage_table(st, steps):
        return {k: round(st[k][0] / steps, 4) for k in ("scale_down", "detect_multi", "describe_all", "laplace_multi",
                                                         "find_points_multi", "compute_orientations",
                                                         "extract_descriptors", "total")}

    def run_single_stream(extractor, imgs, steps, warm=2):
        """`steps` extractions on one stream with per-launch HIP events; returns (ms per step, stage dict)."""
        for _ in range(warm):
            extractor.extract(imgs)
        torch.cuda.synchronize()
        extractor.ctx.timing_enable(True)
        extractor.ctx.timing_reset()
        t1 = time.perf_counter()
        for _ in range(steps):
            extractor.extract(imgs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / steps * 1e3
        st = extractor.ctx.timing_read()
        extractor.ctx.timing_enable(False)
        return ms, st

    def run_pipelined(imgs, steps, warm=None, **param_overrides):
        """`steps` extractions rotated over the E streams exactly as in the timed region (no gather); returns ms/step.
        param_overrides are set on every extractor for the duration."""
        warm = E if warm is None else warm
        saved = [{k: getattr(x.params, k) for k in param_overrides} for x in exs]
        for x in exs:
            x.params.concurrent_batches = E
            for k, v in param_overrides.items():
                setattr(x.params, k, v)
        try:
            for _ in range(warm):
                pipe.submit(imgs)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                pipe.submit(imgs)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / steps * 1e3
        finally:
            for x, sv in zip(exs, saved):
                for k, v in sv.items():
                    setattr(x.params, k, v)

    class leg_guard:
        """An extra leg never costs the line: an exception inside it is recorded under `leg_errors` and the extractors'
        parameters are put back to the timed region's."""

        def __init__(self, name):
            self.name = name

        def __enter__(self):
            return self

        def __exit__(self, et, ev, tb):
            if et is None or not issubclass(et, Exception):
                return False
            out.setdefault("leg_errors", {})[self.name] = "%s: %s" % (et.__name__, ev)
            print("bench.py: leg %s failed: %s: %s" % (self.name, et.__name__, ev), file=sys.stderr)
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            for x in exs:
                x.params.concurrent_batches = 1 if x is ex else E
                x.params.init_blur = args.init_blur
                x.params.fused_detect = 0 if args.two_stage else 1
            return True

    if rank == 0 and legs:
        torch.cuda.synchronize()
        if "repeat" in legs and not use_dist:
            reps_in_order = [run_pipelined(d_imgs, K, warm=0) for _ in range(4)]
            reps = sorted(reps_in_order)
            allr = sorted(reps + [elapsed / K * 1e3])
            out["ms_per_step_spread"] = {"min": round(allr[0], 4), "median": round(allr[len(allr) // 2], 4),
                                         "max": round(allr[-1], 4), "regions": len(allr),
                                         "in_order": [round(elapsed / K * 1e3, 4)] + [round(r, 4) for r in reps_in_order],
                                         "note": "the timed region (`ms_per_step`) and 4 repeats of it, K steps each"}
            time.sleep(0.05)  # what a region costs that starts from an idle device (no pre-flight, no warm-up)
            idle_ms = run_pipelined(d_imgs, K, warm=0)
            out["ms_per_step_spread"]["ms_per_step_from_idle"] = round(idle_ms, 4)
            # beside `value`, at the top level: the same K steps started from an idle device, no pre-flight, no warm-up
            out["ms_per_step_from_idle"] = round(idle_ms, 4)
            out["value_from_idle_mpix_per_s"] = round(total_pix / idle_ms / 1e3, 1)
            # same box, same images, the reference's order (ScaleDown chain first, coarsest octave searched first):
            # what the pyramid-in-detection sequence is worth here
            if args.pyramid_in_detect == -1:
                with leg_guard("pyramid_policy_ab"):
                    ab = {}
                    for pol, name in ((0, "scale_down_chain_first"), (-1, "pyramid_in_detect (default)")):
                        for x in exs:
                            x.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, pol)
                        r = sorted(run_pipelined(d_imgs, K) for _ in range(3))
                        ab[name] = {"ms_per_step_median_of_3": round(r[1], 4), "Mpix_per_s": round(total_pix / r[1] / 1e3, 1)}
                    out["pyramid_policy_ab"] = ab
                for x in exs:
                    x.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, -1)
        ex.params.concurrent_batches = 1  # the single-stream legs below run one batch at a time on one stream
        if stage_overlapped is not None and E > 1:
            out["timed_region_kernel_spans_ms_per_step"] = dict(
                stage_table(stage_overlapped, K), ms_per_step_of_this_repeat=round(spans_ms_per_step, 4),
                note="a repeat of the timed region WITH the stage timers on (the timed region itself runs without them)")

        # ---- single-stream leg: per-stage table, VALU rooflines of the two kernels that own the step ----
        stage = None
        if "single" in legs:
            # the TIMED REGION's launch sequence (every detection writes the next octave: five detect_fused_kernel launches,
            # the join, the description) on ONE stream, with the chunk heights of a caller that has the GPU to itself
            # (concurrent_batches = 1: the timed region's tall chunks only pay with other batches filling the tails -- on one
            # stream they cost 1.07 against 0.83 ms).  profiles/valu.json counts exactly these launches (tools/profile_gpu.sh,
            # the one-stream PMC pass).  What a lone caller gets by DEFAULT (octave 1 from octave 0's detection, one launch for
            # the coarser octaves) is measured right below as lone_caller_ms_per_step
            ex.params.concurrent_batches = 1
            if args.pyramid_in_detect == -1:
                ex.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, 2)
            single_ms, stage = run_single_stream(ex, d_imgs, K)
            if args.pyramid_in_detect == -1:
                ex.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, -1)
            out["stage_ms_per_step"] = stage_table(stage, K)
            # a lone caller: one batch at a time on one stream, no stage timers -- the driver then runs octave 0's
            # detection on the context's second stream beside the ScaleDown chain and the coarser octaves
            lone_steps = 0 if args.profile_run else K  # (not under the profiler: its per-kernel averages are per launch)

            def lone(steps):
                for _ in range(2 if steps else 0):
                    ex.extract(d_imgs)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(steps):
                    ex.extract(d_imgs)
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / max(1, steps) * 1e3

            lone_ms = lone(lone_steps)  # the default policy: nothing forks
            forks0 = ex.ctx.forks()
            if lone_steps:  # the side stream is opt-in (nothing in the timed region or any other leg uses it); with it
                # octave 0 cannot hand octave 1 to the coarser detections, so the ScaleDown chain runs beside it
                ex.ctx.set_policy(capi.POLICY_SIDE_STREAM, 2)  # after the probe: four other streams are in use here
            lone_forked_ms = lone(lone_steps)
            ex.ctx.set_policy(capi.POLICY_SIDE_STREAM, 0)
            out["single_stream_leg"] = {
                "ms_per_step": round(single_ms, 4),
                "lone_caller_ms_per_step": round(lone_ms, 4) if lone_steps else None,
                "lone_caller_side_stream_ms_per_step": round(lone_forked_ms, 4) if lone_steps else None,
                "lone_caller_forked_steps": max(0, ex.ctx.forks() - forks0 - 2),
                "note": "the timed region alternates steps over %d streams; stage_ms_per_step, the VALU rooflines and "
                        "pyramid_mpix_per_s are measured on the same steps -- the same launch sequence (every detection "
                        "writes the next octave), chunk heights of concurrent_batches = 1 -- run on one stream (HIP events per launch), "
                        "where kernel spans do not overlap.  lone_caller_ms_per_step: the same calls without the stage "
                        "timers and with concurrent_batches = 1 -- what a caller that keeps ONE batch in flight gets by "
                        "default (octave 1 from octave 0's detection, one launch for the coarser octaves, short chunks); lone_caller_side_stream_ms_per_step: with CUSIFT_POLICY_SIDE_STREAM = 2 (octave 0's "
                        "detection on the context's second stream beside the ScaleDown chain and the coarser octaves)" % E}
            sd_ms = stage["scale_down"][0]
            det_ms, det_n = stage["detect_multi"]
            if det_n > 0:
                out["pyramid_mpix_per_s"] = round(B * w * h / ((sd_ms + det_ms) / K * 1e-3) / 1e6, 1)
            out["scale_down_GBps"] = round(down_b / (sd_ms / K * 1e-3) / 1e9, 1) if sd_ms > 0 else None

            def valu_roofline(kernel, stage_key, note):
                ms, n = stage[stage_key]
                info = valu.get(kernel)
                if n == 0 or ms <= 0 or not info:
                    return None
                # FMA-equivalent flop: every VALU lane-operation priced as one FMA (2 flop) -- the pricing of the
                # 157.3 TFLOP/s peak (32 lanes x 2 flop per SIMD-clock), so frac = vector issue slots used
                insts_per_step = info["valu_wave_insts_per_launch"] * (n / K)
                ach = insts_per_step * 64 * 2 / (ms / K * 1e-3) / 1e12
                r = {"kernel": kernel, "bound": "valu", "achieved": round(ach, 2), "peak": FP32_VALU_PEAK_TF,
                     "unit": "TFLOP/s", "frac": round(ach / FP32_VALU_PEAK_TF, 4),
                     "valu_wave_insts_per_step": int(insts_per_step), "launches_per_step": n // K,
                     "ms_per_step": round(ms / K, 4),
                     "hbm_traffic_bytes_per_launch": traffic.get(kernel, {}).get("hbm_bytes_per_launch"),
                     "counters_source": "profiles/valu.json, profiles/traffic.json: the builder's rocprofv3 --pmc passes "
                                        "of this command, committed with the kernels they count -- NOT collected in this "
                                        "run (only the times are)",
                     "note": note}
                # The spec peak prices every wave-instruction at 2 cycles per SIMD; only the plain fp32 / integer add,
                # multiply, fma, logic and move forms with no SGPR operand come near it (2.65), every other form --
                # packed, DPP, min / max, compare, select, convert, anything that reads an SGPR -- costs 4.2 and a
                # transcendental 8.2 (tools/microbench/valu_rate.hip, profiles/r03/valu_rate_forms.txt).  Issue bound =
                # PMC instruction count x the mix-weighted cycles per instruction (static mix of the hot loop blocks from
                # the ISA, tools/isa_mix.py) / (1024 SIMDs x 2.4 GHz): the time the SIMDs need just to ISSUE the kernel
                # -- at the best the hardware does per class (eight waves per SIMD), and at what the classes cost with
                # the kernel's own number of resident waves.
                mix = isa_mix.get(kernel)
                if mix and "cycles_per_instruction_at_occupancy" in mix:
                    keys = ("cycles_per_instruction_mix_weighted", "cycles_per_instruction_at_occupancy")
                    cpi = [mix[k] for k in keys]
                    detail = {"mix": mix["mix"]}
                    ana = mix.get("analysis")
                    if ana:  # fused detection: every wave-row runs the blur blocks, a fraction p of them the analysis
                        p_pass = 1.0 - PRETEST_SKIP_HEADLINE
                        nb, na = mix["instructions_per_row_step"], ana["instructions_per_row_step"]
                        cpi = [(nb * mix[k] + p_pass * na * ana[k]) / (nb + p_pass * na) for k in keys]
                        detail = {"blur_blocks": {"instructions_per_row": nb, "mix": mix["mix"],
                                                  "cycles_per_instruction": mix[keys[0]],
                                                  "cycles_per_instruction_at_occupancy": mix[keys[1]]},
                                  "analysis_blocks": {"instructions_per_row": na, "mix": ana["mix"],
                                                      "cycles_per_instruction": ana[keys[0]],
                                                      "cycles_per_instruction_at_occupancy": ana[keys[1]],
                                                      "rows_that_run_them": round(p_pass, 3)}}
                    bound_ms = [insts_per_step * c / (1024 * 2.4e9) * 1e3 for c in cpi]
                    r["issue_bound"] = dict(detail, cycles_per_wave_instruction=round(cpi[0], 3),
                                            cycles_per_wave_instruction_at_occupancy=round(cpi[1], 3),
                                            waves_per_simd=mix["waves_per_simd"],
                                            bound_ms_per_step=round(bound_ms[0], 4),
                                            frac_of_issue_bound=round(bound_ms[0] / (ms / K), 4),
                                            model_ms_per_step_at_own_occupancy=round(bound_ms[1], 4),
                                            note="bound_ms / measured ms: 1.0 = the vector pipes issue back to back.  "
                                                 "frac_of_issue_bound prices each class at the best the SIMD does for "
                                                 "it (eight resident waves): a BOUND.  model_ms_per_step_at_own_occupancy "
                                                 "prices the classes at what independent chains cost with this kernel's "
                                                 "resident waves (two waves: a slow-class instruction 4.6-5.7 cycles by "
                                                 "run, 5.1 used) -- a MODEL, not a bound: round 4 printed its ratio to the "
                                                 "measurement as `frac_at_own_occupancy` and it came out at 1.06 for the "
                                                 "description kernel (its LDS and memory instructions interleave with the "
                                                 "vector ones better than the microbenchmark's chains do); the field is "
                                                 "gone.  The mix is a static estimate (profiles/isa_mix.json)")
                return r

            rk = []
            r = valu_roofline("detect_fused_kernel", "detect_multi",
                              "achieved = PMC SQ_INSTS_VALU (profiles/valu.json, same command) x 64 lanes x 2 flop / "
                              "HIP-event time of this run")
            if r:
                rk.append(r)
            r = valu_roofline("describe_all_kernel", "describe_all", "as above; %d keypoints per step"
                              % local_kp)
            if r:
                r["valu_wave_insts_per_keypoint"] = round(r["valu_wave_insts_per_step"] / max(1, local_kp), 1)
                rk.append(r)
            if rk:
                out["roofline_kernels"] = rk

        # ---- two-stage leg: the blur+DoG kernel the north star names ----
        if "two_stage" in legs and ex.params.fused_detect:
            ex.params.fused_detect = 0
            two_ms, stage2 = run_single_stream(ex, d_imgs, K)
            ex.params.fused_detect = 1
            lap_ms, lap_n = stage2["laplace_multi"]
            if lap_n > 0 and lap_ms > 0:
                # per launch: mean algorithmic bytes / mean HIP-event duration over the launches (5 octaves x K steps)
                achieved = (blur_b * K / lap_n) / (lap_ms * 1e-3 / lap_n) / 1e9
                out["roofline"] = {
                    "kernel": "laplace_multi_fast_kernel (8 blurs + 7 DoG planes, 32 B/px algorithmic)",
                    "bound": "hbm",
                    "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic.get("laplace_multi_fast_kernel", {}).get("hbm_bytes_per_launch"),
                    "traffic_source": "profiles/traffic.json (the builder's FETCH_SIZE / WRITE_SIZE passes of this command, "
                                      "gfx950 corrections applied; committed, not collected in this run)",
                    "algorithmic_bytes_per_launch": int(blur_b * K / lap_n),
                    "avg_launch_ms": round(lap_ms / lap_n, 5),
                    "launches": lap_n,
                    "note": "measured in the two-stage leg of this run (same inputs, HIP events on the launching "
                            "stream); the timed region itself uses the fused kernel, whose roofline is VALU "
                            "(roofline_kernels)",
                }
                # the octave-0 launch on its own (3/4 of the bytes): the same kernel through the stage entry point,
                # DoG planes of the whole batch in a buffer of their own
                try:
                    if args.profile_run:
                        raise RuntimeError("skipped (--profile-run)")
                    dog0 = torch.empty((B, 7, h, ex.pitch), dtype=torch.float32, device=dev)
                    ex.ctx.timing_enable(True)
                    for rep in range(2 + max(3, K // 2)):
                        if rep == 2:
                            torch.cuda.synchronize()
                            ex.ctx.timing_reset()
                        ex.ctx.laplace_multi(d_imgs.data_ptr(), w, h, ex.pitch, args.init_blur, dog0.data_ptr(),
                                             n_images=B, img_stride=h * ex.pitch, dog_stride=7 * h * ex.pitch)
                    torch.cuda.synchronize()
                    l0_ms, l0_n = ex.ctx.timing_read()["laplace_multi"]
                    ex.ctx.timing_enable(False)
                    del dog0
                    b0 = 32.0 * w * h * B
                    out["roofline"]["octave0_launch"] = {
                        "algorithmic_bytes": int(b0), "avg_launch_ms": round(l0_ms / l0_n, 5),
                        "achieved": round(b0 / (l0_ms / l0_n * 1e-3) / 1e9, 1),
                        "frac": round(b0 / (l0_ms / l0_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "note": "the largest launch alone; `achieved` above averages it with the four smaller octaves"}
                except Exception as e:  # noqa: BLE001 -- an extra, never the reason to lose the line
                    out["roofline"]["octave0_launch"] = {"error": "%s: %s" % (type(e).__name__, e)}
            fp_ms = stage2["find_points_multi"][0]
            out["two_stage_leg"] = {"ms_per_step": round(two_ms, 4), "stage_ms_per_step": stage_table(stage2, K),
                                    "find_points_GBps": round(find_b / (fp_ms / K * 1e-3) / 1e9, 1) if fp_ms > 0 else None,
                                    "find_points_traffic_bytes_per_launch":
                                        traffic.get("find_points_fast_kernel", {}).get("hbm_bytes_per_launch")}
        elif "two_stage" in legs and stage_overlapped is not None:  # --two-stage: the timed region itself
            lap_ms, lap_n = stage_overlapped["laplace_multi"]
            if lap_n:
                achieved = (blur_b * K / lap_n) / (lap_ms * 1e-3 / lap_n) / 1e9
                out["roofline"] = {"kernel": "laplace_multi_fast_kernel", "bound": "hbm", "achieved": round(achieved, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                                   "traffic": traffic.get("laplace_multi_fast_kernel", {}).get("hbm_bytes_per_launch"),
                                   "note": "measured over the timed region (kernel spans of %d streams)" % E}

        # ---- host-visible leg: SiftData in pinned host memory, copies overlapped with the next step ----
        if "host" in legs:
            with leg_guard("host"):
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

        # ---- host-to-host legs: what a caller of the reference's entry point (host image in, host SiftData out) gets ----
        if "host_in" in legs:
            with leg_guard("host_in"):
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

        # ---- content legs ----
        if "content" in legs:
            with leg_guard("content"):
                cl = {}
                cl["tile_preblurred (the timed workload)"] = content_stats(torch, capi, ex, d_imgs, None, w, h, B, args, K)
                def pipelined_rate(d, imgs):
                    ex.params.concurrent_batches = E
                    ms = run_pipelined(imgs, max(8, K // 2))
                    ex.params.concurrent_batches = 1
                    d["ms_per_step_pipelined"] = round(ms, 4)
                    d["Mpix_per_s_pipelined"] = round(B * w * h / (ms * 1e-3) / 1e6, 1)
                    d["keypoints_per_s_pipelined"] = round(d["keypoints_per_step"] / (ms * 1e-3), 1)
                    return d["Mpix_per_s_pipelined"]

                raw = ex.images_from_numpy(make_images(lambda s: synth.tile(s, w, h, 0.0), seeds))
                name = "tile_raw (SURVEY 8d primary generator as written: no pre-blur; initBlur=%.1f still declared)" % args.init_blur
                cl[name] = content_stats(torch, capi, ex, raw, run_single_stream, w, h, B, args, max(4, K // 2))
                out["value_tile_raw_mpix_per_s"] = pipelined_rate(cl[name], raw)
                del raw
                blob = ex.images_from_numpy(make_images(lambda s: synth.blobs(s, w, h), seeds))
                name = "blobs (SURVEY 8d secondary generator)"
                cl[name] = content_stats(torch, capi, ex, blob, run_single_stream, w, h, B, args, max(4, K // 2))
                out["value_blobs_mpix_per_s"] = pipelined_rate(cl[name], blob)
                del blob
                if stage is not None:
                    cl["tile_preblurred (the timed workload)"].update(
                        {"ms_per_step_single_stream": out["single_stream_leg"]["ms_per_step"],
                         "keypoints_per_step": local_kp})
                out["content_legs"] = cl

        # ---- initBlur = 0 leg: the timed images with no blur declared (test/detector.cpp:43) ----
        if "initblur0" in legs:
            with leg_guard("initblur0"):
                saved_blur = ex.params.init_blur
                ex.params.init_blur = 0.0
                i_ms, i_st = run_single_stream(ex, d_imgs, max(4, K // 2))
                i_kp = int(ex.valid_counts().sum().item())
                raw_cnt = torch.clamp(ex.counts, min=0)
                ex.params.init_blur = saved_blur
                ex.params.concurrent_batches = E
                p_ms = run_pipelined(d_imgs, max(8, K // 2), init_blur=0.0)
                ex.params.concurrent_batches = 1
                n_steps = max(4, K // 2)
                out["initblur0_leg"] = {
                    "workload": "the timed images, initBlur = 0.0 declared: all 8 levels of octave 0 are filtered (no "
                                "identity pass-through), the detector sees more and finer structure",
                    "ms_per_step_single_stream": round(i_ms, 4), "ms_per_step_pipelined": round(p_ms, 4),
                    "Mpix_per_s_pipelined": round(B * w * h / (p_ms * 1e-3) / 1e6, 1), "keypoints_per_step": i_kp,
                    "keypoints_per_s_pipelined": round(i_kp / (p_ms * 1e-3), 1),
                    "images_saturating_max_pts": int((raw_cnt >= ex.max_pts).sum().item()),
                    "stage_ms_per_step": {k: round(i_st[k][0] / n_steps, 4) for k in ("scale_down", "detect_multi",
                                                                                      "describe_all")}}
                out["value_initblur0_mpix_per_s"] = out["initblur0_leg"]["Mpix_per_s_pipelined"]

        # ---- ragged-width leg ----
        if "ragged" in legs:
            with leg_guard("ragged"):
                rw, rh = 1366, 768
                rex = BatchExtractor(B, rw, rh, **prm_kw)
                rimgs = rex.images_from_numpy(make_images(lambda s: synth.tile(s, rw, rh, args.init_blur), seeds))
                r_ms, r_st = run_single_stream(rex, rimgs, max(4, K // 2))
                rate = B * rw * rh / (r_ms * 1e-3) / 1e6
                leg = {"workload": "%d x %dx%d (octave widths 1366, 683, 341, 170, 85: none a multiple of 4)" % (B, rw, rh),
                       "ms_per_step_single_stream": round(r_ms, 4), "Mpix_per_s_single_stream": round(rate, 1),
                       "stage_ms_per_step": {k: round(r_st[k][0] / max(4, K // 2), 4)
                                             for k in ("scale_down", "detect_multi", "describe_all")},
                       "detect_launches_fused": r_st["detect_multi"][1], "laplace_launches": r_st["laplace_multi"][1],
                       "keypoints_per_step": int(rex.valid_counts().sum().item())}
                if "single_stream_leg" in out:
                    base = B * w * h / (out["single_stream_leg"]["ms_per_step"] * 1e-3) / 1e6
                    leg["per_pixel_rate_vs_1080p"] = round(rate / base, 3)
                out["ragged_width_leg"] = leg
                rex.close()
                del rimgs

        # ---- matcher leg (the first caller after the path): fp32 MFMA bound ----
        if "match" in legs:
            with leg_guard("match"):
                out["match_leg"] = match_leg(capi, ex.ctx, 16384)

        # the headline is ONE content; the number to carry is the range over the survey's generators
        rates = {"tile_preblurred (timed region)": out["value"]}
        for key, label in (("value_blobs_mpix_per_s", "blobs"), ("value_tile_raw_mpix_per_s", "tile_raw (every image "
                           "saturates maxPts)"), ("value_initblur0_mpix_per_s", "tile_preblurred, initBlur=0")):
            if key in out:
                rates[label] = out[key]
        if len(rates) > 1:
            out["value_range_mpix_per_s"] = [min(rates.values()), max(rates.values())]
            out["value_by_content_mpix_per_s"] = rates

        # `roofline` is the blur + DoG exhibit the north star gates (a kernel the timed region does not launch); the timed
        # step's own dominant kernel and ITS bound ride inside it, so that whoever copies `roofline` has both
        if "roofline" in out and out.get("roofline_kernels"):
            dom = max(out["roofline_kernels"], key=lambda r: r["ms_per_step"])
            ib = dom.get("issue_bound", {})
            out["roofline"]["timed_region"] = {
                "kernel": dom["kernel"], "bound": "valu", "achieved": dom["achieved"], "peak": dom["peak"],
                "unit": dom["unit"], "frac": dom["frac"], "ms_per_step_single_stream": dom["ms_per_step"],
                "launches_per_step": dom["launches_per_step"], "frac_of_issue_bound": ib.get("frac_of_issue_bound"),
                "share_of_single_stream_step": round(dom["ms_per_step"] / out["single_stream_leg"]["ms_per_step"], 3)
                if out.get("single_stream_leg") else None,
                "note": "the kernel that owns the timed step (fused blur + DoG + extrema + refinement + the next octave's "
                        "image): vector-issue bound at two waves per SIMD, moves 4 B per pixel of HBM traffic -- its HBM "
                        "roofline fraction is meaningless by design (DoG planes never leave the chip)"}

        if "cpu" in legs:
            with leg_guard("cpu"):
                out["cpu_baseline"] = cpu_baseline(w, h, dict(prm_kw), args.init_blur, args.cpu_seconds)
                if world > 1:
                    out["cpu_baseline"]["note"] = "timed on rank 0's host share after the timed region (other ranks idle)"

    if world > 1:
        dist.barrier()  # the other ranks wait here while rank 0 runs its legs: communicators are torn down together
    for x in exs:
        x.close()
    if comm is not None:
        comm.close()
    if side_ctx is not None:
        side_ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


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


def content_stats(torch, capi, ex, d_imgs, run_single_stream, w, h, B, args, steps):
    """Keypoints per step and the fraction of octave-0 wave-rows (240 columns x 1 row, the fused kernel's unit) in
    which no DoG centre of the 5 searchable scales exceeds the threshold -- the rows the pre-test skips (measured from
    the DoG planes of image 0 through the two-stage entry point); plus the single-stream rate on this content."""
    out = {}
    p = ex.pitch
    dog = torch.empty((7, h, p), dtype=torch.float32, device=d_imgs.device)
    ex.ctx.laplace_multi(d_imgs.data_ptr(), w, h, p, args.init_blur, dog.data_ptr())
    torch.cuda.synchronize()
    big = (dog[1:6, 1:h - 1, :w].abs() > args.thresh).any(dim=0)   # [h-2, w]: any scale above threshold, centre rows
    big[:, 0] = False  # border columns are never centres
    big[:, w - 1] = False
    strips = -(-w // 240)  # the kernel's strips: columns [240 s, 240 s + 240)
    pad = torch.zeros((big.shape[0], strips * 240), dtype=torch.bool, device=big.device)
    pad[:, :w] = big
    rows_with = pad.view(big.shape[0], strips, 240).any(dim=2)
    out["pretest_skip_frac_octave0"] = round(1.0 - float(rows_with.float().mean().item()), 4)
    out["pixels_above_thresh_frac_octave0"] = round(float(big.float().mean().item()), 5)
    if run_single_stream is not None:
        ms, st = run_single_stream(ex, d_imgs, steps)
        out["ms_per_step_single_stream"] = round(ms, 4)
        out["Mpix_per_s_single_stream"] = round(B * w * h / (ms * 1e-3) / 1e6, 1)
        out["keypoints_per_step"] = int(ex.valid_counts().sum().item())
        out["stage_ms_per_step"] = {k: round(st[k][0] / steps, 4) for k in ("scale_down", "detect_multi", "describe_all")}
        raw = torch.clamp(ex.counts, min=0)
        out["images_saturating_max_pts"] = int((raw >= ex.max_pts).sum().item())
    return out


if __name__ == "__main__":
    main()
