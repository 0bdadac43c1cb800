#!/usr/bin/env python3
"""bench.py -- the headline benchmark of the MI355X-native SIFT extraction path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
(the first form with N > 1 starts the second one itself, as a child process, before anything touches a GPU)

Workload (BASELINE.json configs[2]/[3]): a batch of 64 synthetic 1920x1080 8-bit-valued images per GPU
(seeded `tile` images pre-blurred to sigma 1.0), 5 octaves, initBlur=1.0, thresh=3.0, edge=10.
One "step" = one pass of the whole hot path over that batch: ScaleDown pyramid (written by the detection launches
themselves: CUSIFT_POLICY_PYRAMID_IN_DETECT, --pyramid-in-detect), 8 blurs + 7 DoG per octave, extrema + refinement,
orientation, 128-D descriptors -- SiftData left in HBM; with N>1 ranks the step ends with the RCCL all-gatherv of
SiftData (C ABI: cusift_allgatherv_*, ncclAllGather of the counts + one ncclGroup of ncclSend/ncclRecv; 540-byte
trimmed records expanded on arrival to 588-byte SiftPoint records unless --gather-exact) so every rank holds all
N*64 images' keypoints.  Inputs are resident in HBM before the timed region.  Weak scaling: 64 images per GPU at every N.
Consecutive steps rotate over --streams HIP streams (default 4, one extractor each), so that the launch tails of one
batch overlap the kernels of the others; every step is still one complete pass over one batch, and the timed region is
bracketed by barrier + device-wide synchronisation on both sides.

Measurement protocol (bench_legs/timed.py; fixed, identical on every rank): set-up check -> W warm-up + K timed steps
exactly as the contract words them (`value_no_preflight_mpix_per_s`) -> --preflight rounds of untimed load (28 steps)
-> W warm-up + K timed steps at the device's steady clocks (`value`).  --preflight 0: the W + K steps alone.

Prints ONE JSON line on rank 0 (see the driver contract in the task description).  `value` comes from the timed
region only.  Everything else on the line is measured in separate, labelled legs after it (bench_legs/*.py, same inputs
unless the leg says otherwise); `--legs` selects them:
  single     the same steps on ONE stream with HIP events per launch: per-stage ms (kernel spans do not overlap)
  two_stage  the reference's LaplaceMulti -> DoG in HBM -> FindPointsMulti pipeline: `roofline` (blur+DoG kernel,
             algorithmic bytes / HIP-event time vs 8 TB/s, the north-star gate)
  host       SiftData made host-visible: packed records copied to pinned memory on a copy stream, overlapped with
             the next step (`keypoints_per_s_host_visible`; SURVEY.md section 8d's end-to-end definition)
  host_in    HOST to HOST: the batch starts as 8-bit pixels (and as float32) in pinned host memory, is uploaded every
             step, converted on the device, extracted, and its SiftData copied back (`host_to_host`)
  repeat     the timed region four more times: min / median / max ms per step; the same region from an idle device
  content    the same pipeline on other image content (`blobs`, un-pre-blurred `tile`): how much of the rate is the images
  initblur0  the timed images with initBlur = 0.0 declared (the only value the reference's own test uses)
  ragged     64 x 1366x768 (no octave width is a multiple of 4): per-pixel rate next to 1080p's
  configs    `config_legs`: BASELINE configs[0] (640x480 fixture, 3 octaves), configs[1] (ONE 1080p frame: latency,
             back-to-back rate, host float in -> host SiftData out), configs[4] (ONE 8192x8192 image: whole on one GPU,
             8 virtual ranks, and with N > 1 strip-tiled over the live ranks) + `tiled_model`, its committed prediction
  match      MatchSiftData (section 8 row f1) on 16384 x 16384 descriptors: TFLOP/s vs the fp32 MFMA peak
  cpu        `cpu_baseline`: the CPU oracle on this box's host cores over a bounded sample; OpenCV if importable
"""
import argparse
import json
import os
import sys

# ROCr reads its flags once, at hsa_init -- i.e. at the first HIP call of the process -- so this has to be in the
# environment before torch (or libcusift_amd.so) touches the GPU: the host driver only supports dmabuf IPC, and
# without it RCCL's cross-process buffer sharing fails with `hipIpcGetMemHandle: invalid argument`.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4).  The benchmark keeps four
# extraction streams busy and, with N > 1, a fifth for the exchange: on four queues the exchange stream shares a queue
# with an extraction stream, and its waits (for the producer's event, for the peers inside RCCL) block the kernels
# queued behind them -- measured at one rank (--force-gather --no-self-p2p): 1.325 ms per step on 4 queues, 1.233 on 8,
# against 1.157 without any exchange.  Read at HIP initialisation, so it has to be set here.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# (no torch, no HIP: these modules are plain Python -- the spawning parent imports them too)
from bench_legs.common import (ALL_LEGS, FP32_VALU_PEAK_TF, HBM_PEAK_GBS, Run, algorithmic_bytes,  # noqa: E402,F401
                               load_profile_json, octave_dims, usable_cpus)
from bench_legs.launch import dry_launch, spawn_ranks  # noqa: E402
from bench_legs.models import XGMI_LINK_GBPS_PER_DIRECTION, gather_model, tiled_model  # noqa: E402,F401


def parse_args(argv=None):
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
                         "pipelining callers from 2 million pixels per call: every detection writes the next octave's image, no ScaleDown "
                         "launches; lone callers from 64 million: octave 0's detection only), 0 the "
                         "ScaleDown chain first (the reference's order, cuSIFT.cu:175-192), 1 octave 0 only, 2 every octave")
    ap.add_argument("--preflight", type=int, default=7,
                    help="rounds of untimed load between the literal W + K region and the timed one, each one batch per "
                         "extractor output slot (7 rounds = 28 steps with four streams): a FIXED number, the same on every "
                         "rank -- after any idle gap the first ~20 ms of load run 5-12 %% slow (tools/probe_rampup.py) and W "
                         "= 5 steps are 5 ms.  0: no set-up check, no literal region, no pre-flight -- `value` is then the "
                         "contract's W + K steps alone")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the launch only: ranks rendezvous over gloo on the CPU, shard the batch, barrier, "
                         "reduce a time and rank 0 prints a line -- no GPU, no extraction (tests the --gpus N spawn)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
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

    from bench_legs import configs, content, cpu, host, match, repeat, single, timed, two_stage
    from cusift_amd import capi, synth

    # Rank 0 prints exactly ONE line on stdout.  Libraries write there too (RCCL prints a version banner on
    # communicator creation), so from here on file descriptor 1 points at stderr and the JSON line goes to a
    # duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    R = Run(args)
    R.torch, R.dist, R.capi, R.synth, R.legs = torch, dist, capi, synth, legs
    R.rank = rank = int(os.environ.get("RANK", "0"))
    R.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    capi.lib()  # fail loudly if the HIP extension is missing -- there is no fallback
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    local_rank %= max(1, torch.cuda.device_count())  # rehearsals with more ranks than GPUs share devices (RCCL permitting)
    torch.cuda.set_device(local_rank)
    R.local_rank = local_rank
    R.dev = torch.device("cuda", local_rank)
    R.use_dist = world > 1 or args.force_gather
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch.distributed carries the communicator's 128-byte id, the barrier and the max-over-ranks of the time;
        # the SiftData exchange itself is the C ABI's (RCCL called from libcusift_amd.so)
        dist.init_process_group("nccl", device_id=R.dev, rank=rank, world_size=world)
    R.w, R.h, R.B = args.width, args.height, args.batch
    R.prm_kw = dict(num_octaves=args.octaves, init_blur=args.init_blur, peak_thresh=args.thresh, edge_thresh=10.0,
                    lowest_scale=0.0, subsampling=1.0, max_pts=args.max_pts, tex_frac_bits=8)

    timed.setup(R)
    timed.run(R)  # `value`; R.out exists on rank 0 from here on

    # ================================================================================================================
    # Extra legs (rank 0's GPU only; not part of `value`).  With N > 1 the other ranks wait at the final barrier.
    # ================================================================================================================
    R.blur_b, R.down_b, R.find_b = algorithmic_bytes(R.w, R.h, args.octaves, R.B)
    R.traffic = load_profile_json("traffic.json")
    R.valu = load_profile_json("valu.json")
    R.isa_mix = load_profile_json("isa_mix.json")
    out, ex, E = R.out, R.ex, R.E
    if rank == 0 and legs:
        torch.cuda.synchronize()
        if "repeat" in legs and not R.use_dist:
            repeat.run(R)
        ex.params.concurrent_batches = 1  # the single-stream legs below run one batch at a time on one stream
        if R.stage_overlapped is not None and E > 1:
            out["timed_region_kernel_spans_ms_per_step"] = dict(
                R.stage_table(R.stage_overlapped, args.steps), ms_per_step_of_this_repeat=round(R.spans_ms_per_step, 4),
                note="a repeat of the timed region WITH the stage timers on (the timed region itself runs without them)")
        if "single" in legs:
            single.run(R)
        if "two_stage" in legs:
            two_stage.run(R)
        if "host" in legs:
            with R.leg_guard("host"):
                host.run_host(R)
        if "host_in" in legs:
            with R.leg_guard("host_in"):
                host.run_host_in(R)
        if "content" in legs:
            with R.leg_guard("content"):
                content.run_content(R)
        if "initblur0" in legs:
            with R.leg_guard("initblur0"):
                content.run_initblur0(R)
        if "ragged" in legs:
            with R.leg_guard("ragged"):
                content.run_ragged(R)
        if "configs" in legs:
            with R.leg_guard("configs"):
                configs.run_rank0(R)
        if "match" in legs:
            with R.leg_guard("match"):
                out["match_leg"] = match.match_leg(capi, ex.ctx, 16384)

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
            with R.leg_guard("cpu"):
                out["cpu_baseline"] = cpu.cpu_baseline(R.w, R.h, dict(R.prm_kw), args.init_blur, args.cpu_seconds)
                if world > 1:
                    out["cpu_baseline"]["note"] = "timed on rank 0's host share after the timed region (other ranks idle)"
                c0 = out.get("config_legs", {}).get("configs[0]")
                out["cpu_baseline"]["configs0"] = cpu.configs0_cpu(c0["keypoints"] if c0 else None)

    # BASELINE configs[4] over the ranks that are up (N > 1; one rank with --force-gather rehearses it over RCCL's self send
    # / recv): EVERY rank takes part, after rank 0's own legs -- the line is complete but for this leg -- and under a
    # watchdog: RCCL has never been entered with more than one rank from the build box, and a leg must never cost the
    # line.  If the leg (or the teardown's barriers) has not finished in time, rank 0 prints the line as it stands, with
    # the fact on record, and every rank leaves through os._exit.
    printed = []

    def emit():
        if rank == 0 and not printed:
            printed.append(True)
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())

    def watchdog(seconds, what):
        import threading

        def expire():
            if rank == 0:
                out.setdefault("leg_errors", {})[what] = "not finished after %d s: the line was printed without it" % seconds
                emit()
            os._exit(0)

        t = threading.Timer(seconds, expire)
        t.daemon = True
        t.start()
        return t

    if world > 1:
        dist.barrier()  # the other ranks waited here while rank 0 ran its legs (no watchdog: the legs take what they take)
    if R.use_dist and "configs" in legs:
        dog = watchdog(240, "configs[4] tiled over the ranks")
        tiled = None
        try:
            tiled = configs.tiled_8192_all_ranks(R)
        except Exception as e:  # noqa: BLE001 -- (a rank that fails alone leaves the others to the watchdog)
            tiled = {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None
        dog.cancel()
        if rank == 0:
            configs.merge_tiled(R, tiled)
    emit()
    dog = watchdog(120, "teardown")
    timed.teardown(R)
    dog.cancel()
    os.close(json_fd)


if __name__ == "__main__":
    main()
