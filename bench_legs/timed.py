"""The timed region of bench.py: extractors, inputs resident in HBM, the N > 1 exchange, the measurement protocol.

Protocol (identical on every rank, nothing in it adapts to what it measures):
  1. set-up check    every extractor runs the batch once per output slot; all must report identical keypoint counts
  2. literal region  W warm-up steps, barrier + synchronize, K timed steps, barrier + synchronize -- the driver contract's
                     wording with nothing added: `value_no_preflight_mpix_per_s`.  On a device that idled through the
                     process's start-up the first ~20 ms of load run 5-12 % slow (tools/probe_rampup.py), so with the driver's
                     W = 5 (5 ms) this region measures the clock ramp as much as the kernels.
  3. pre-flight      --preflight rounds (default 7) of one batch per extractor output slot: a FIXED number of untimed steps
                     (28 with four streams) that keep the device loaded; config.preflight_steps says how many ran.
  4. timed region    W warm-up steps, barrier + synchronize, K timed steps, barrier + synchronize: `value` -- the same K
                     steps at the device's steady clocks.  --preflight 0 skips 1-3: `value` is then the literal region.
"""
import os
import sys
import time

import numpy as np

from .common import METRIC
from .models import gather_model


def setup(R):
    """Extractors, inputs, and -- with N > 1 or --force-gather -- the side stream and the C-ABI communicator."""
    args, torch, dist, capi, synth = R.args, R.torch, R.dist, R.capi, R.synth
    from cusift_amd.batch import PipelinedExtractor
    from cusift_amd.dist import SiftGatherer, make_comm

    rank, world, local_rank, dev = R.rank, R.world, R.local_rank, R.dev
    w, h, B = R.w, R.h, R.B
    use_dist = R.use_dist
    # One extractor (context + arena + output slots) per stream; step i runs on stream i % E.  A step is still one
    # whole pass of the hot path over one batch -- consecutive steps merely overlap on the device.
    # The command processor has four compute pipes; hardware queues 1, 5, 9 ... share the first.  A fifth busy stream
    # (the exchange) lands on a pipe that an extraction stream uses and the two queues' packets wait for each other:
    # measured at one rank (--force-gather, self send/recv): 4 + 1 streams 1.32-1.33 ms per step, 3 + 1 streams 1.29
    # (without the exchange four streams win: 1.17 against 1.19).
    R.E = E = args.streams if args.streams > 0 else (3 if use_dist else 4)
    R.n_slots = n_slots = 2 if use_dist else 1  # a slot is read by the pack of its step's gather while the next steps run
    R.pipe = pipe = PipelinedExtractor(B, w, h, n_streams=E, n_slots=n_slots,
                                       fused_detect=0 if args.two_stage else 1, **R.prm_kw)
    R.exs = exs = pipe.extractors
    R.ex = ex = exs[0]
    if args.pyramid_in_detect != -1:
        for x in exs:
            x.ctx.set_policy(capi.POLICY_PYRAMID_IN_DETECT, args.pyramid_in_detect)

    # ---- synthetic inputs, resident in HBM before anything is timed ----
    R.seeds = [1000 + rank * B + i for i in range(B)]
    R.np_imgs = R.make_images(lambda s: synth.tile(s, w, h, args.init_blur), R.seeds)
    R.d_imgs = ex.images_from_numpy(R.np_imgs)

    # N > 1: the all-gatherv of step i runs on a side stream (its own context + communicator).  begin(i) -- counts
    # exchange + the local shard packed into its region -- is enqueued right after step i; finish(i) -- the one host READ
    # of the counts, then the grouped ncclSend/ncclRecv -- after step i + LAG has been enqueued.  The host runs AHEAD of
    # the device (enqueueing a step takes ~0.1 ms, executing it ~1.2), so finish(i) usually finds the counts flag not yet
    # set and spins on it: that wait is the host's throttle, not device idle time -- the device still has LAG steps
    # queued.  config.gather_host_waits counts those finishes and config.gather_host_wait_ms is the time spent in them.
    R.main_stream = torch.cuda.current_stream()
    R.side_stream = torch.cuda.Stream() if use_dist else None
    R.LAG = E
    R.gatherer = R.comm = R.side_ctx = None
    R.gather_impl = None
    R.region_cap = B * args.gather_capacity
    if use_dist:
        # The exchange is the C ABI's (RCCL called from libcusift_amd.so).  On the build box it has met more than one
        # rank only over the in-process test transport (tests/test_multirank_gpu.py; RCCL refuses two ranks per GPU),
        # so a failure to bring the communicator up is not allowed to cost the run: all ranks then agree to fall back to
        # the torch.distributed twin of the same exchange, and the JSON line says which one ran (config.gather_impl).
        err = ""
        try:
            R.side_ctx = capi.Context(local_rank, stream=R.side_stream.cuda_stream)
            R.comm = make_comm(R.side_ctx, self_p2p=(world == 1 and not args.no_self_p2p))
            R.gatherer = SiftGatherer(R.comm, B, args.max_pts, region_cap=R.region_cap, device=dev, n_out=R.LAG + 2,
                                      depth=R.LAG + 1, wire_format="compact" if args.gather_compact else (
                                          "exact" if args.gather_exact else "trimmed"),
                                      expand=not (args.gather_compact or args.gather_exact or args.gather_trimmed))
        except Exception as e:  # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            R.gather_impl = ("C ABI (cusift_allgatherv_*): ncclAllGather of counts + one ncclGroup of ncclSend/ncclRecv, "
                             "finish lags begin by %d steps" % R.LAG)
        else:
            print("bench.py: C-ABI communicator unavailable (%s); using the torch.distributed exchange" % err,
                  file=sys.stderr)
            R.gatherer = None
            R.LAG = 1
            R.gather_impl = "torch.distributed fallback (C ABI communicator failed: %s)" % (err or "on another rank")


def run(R):
    """Steps 1-4 of the protocol; fills R.elapsed (max over ranks), the keypoint totals and -- on rank 0 -- R.out."""
    args, torch, dist, capi = R.args, R.torch, R.dist, R.capi
    from cusift_amd.dist import begin_allgather, finish_allgather

    rank, world, dev = R.rank, R.world, R.dev
    w, h, B, E, n_slots = R.w, R.h, R.B, R.E, R.n_slots
    pipe, exs, ex, d_imgs = R.pipe, R.exs, R.ex, R.d_imgs
    use_dist, gatherer, comm, side_stream, main_stream = R.use_dist, R.gatherer, R.comm, R.side_stream, R.main_stream
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
            if len(pending) > R.LAG:
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

    def region(steps):
        """W warm-up steps, fence, `steps` timed steps, fence: seconds of the timed part on this rank."""
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        return time.perf_counter() - t0

    R.step, R.fence, R.drain = step, fence, drain
    # The timed regions run the PRODUCT: stage timers off (two hipEventRecord per launch, and a driver that pins the
    # per-octave launch sequence while they are on), whatever legs follow -- `--legs none` takes the same path, and
    # config.timed_region_forks / _timers say so.
    torch.cuda.synchronize()
    K = args.steps
    preflight = max(0, args.preflight)
    elapsed_literal = None
    R.preflight_steps = 0
    if preflight:
        # 1. set-up check (four contexts, four arenas, one answer)
        for _ in range(E * n_slots):
            pipe.submit(d_imgs)
        pipe.synchronize()
        ref_counts = exs[0].slots[0][1].clone()
        for x in exs:
            for _, cnt_t in x.slots:
                if not torch.equal(cnt_t, ref_counts):
                    raise SystemExit("bench.py: set-up check: extractors disagree on the keypoint counts of the same batch")
        # 2. the contract's wording with nothing added
        elapsed_literal = region(K)
        # 3. a fixed number of untimed steps, the same on every rank; nothing but enqueueing between here and the
        #    timed region's fence
        R.preflight_steps = preflight * E * n_slots
        for _ in range(R.preflight_steps):
            pipe.submit(d_imgs)
    # 4. the timed region
    forks_before = sum(x.ctx.forks() for x in exs)
    elapsed = region(K)
    R.forks_timed = sum(x.ctx.forks() for x in exs) - forks_before
    gathered = state["gathered"]
    R.gather_waits = (comm.host_waits(), comm.host_wait_ms()) if comm is not None else None
    if R.legs and not use_dist:
        # the kernel-span table of the overlapped streams: a REPEAT of the region with the stage timers on
        for x in exs:
            x.ctx.timing_enable(True)
            x.ctx.timing_reset()
        t1 = time.perf_counter()
        for _ in range(K):
            step()
        fence()
        R.spans_ms_per_step = (time.perf_counter() - t1) / K * 1e3
        for x in exs:  # kernel spans of all streams (with E > 1 they overlap in time: their sum exceeds the wall time)
            t = x.ctx.timing_read()
            R.stage_overlapped = t if R.stage_overlapped is None else {
                k: (R.stage_overlapped[k][0] + t[k][0], R.stage_overlapped[k][1] + t[k][1]) for k in t}
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
        for _ in range(K):
            pipe.submit(d_imgs)
        pipe.synchronize()
        eo = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(eo, op=dist.ReduceOp.MAX)
        extraction_only_ms = float(eo.item()) / K * 1e3
    # max over ranks (and every rank's own time, for the line)
    el = torch.tensor([elapsed, elapsed_literal if elapsed_literal is not None else elapsed], dtype=torch.float64,
                      device=dev)
    per_rank_elapsed = [elapsed]
    if world > 1:
        every = [torch.zeros_like(el) for _ in range(world)]
        dist.all_gather(every, el)
        per_rank_elapsed = [float(t[0].item()) for t in every]
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    R.elapsed = elapsed = float(el[0].item())
    elapsed_literal = float(el[1].item())
    counts = ex.valid_counts()
    R.local_kp = local_kp = int(counts.sum().item())
    kp = torch.tensor([local_kp], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(kp, op=dist.ReduceOp.SUM)
    R.total_kp = total_kp = int(kp.item())
    if use_dist:
        total_gathered = int(np.asarray(gathered[2], dtype=np.int64).sum())
        assert total_gathered == total_kp, (total_gathered, total_kp)

    R.ms_per_step = elapsed / K * 1e3
    R.total_pix = world * B * w * h
    if rank != 0:
        return
    R.out = out = {
        "metric": METRIC,
        "value": round(R.total_pix / (elapsed / K) / 1e6, 2),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": round(R.ms_per_step, 4),
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
            "timed_region_forks": int(R.forks_timed),
            "preflight_steps": R.preflight_steps,
            "preflight_note": "protocol (bench_legs/timed.py), identical on every rank: set-up check (all extractors must "
                              "report identical keypoint counts) -> W + K steps exactly as the contract words it "
                              "(value_no_preflight_mpix_per_s) -> a FIXED number of untimed steps (preflight_steps) -> W "
                              "warm-up + K timed steps at steady clocks (`value`); --preflight 0 runs the W + K steps alone",
            "pyramid_in_detect": ex.ctx.get_policy(capi.POLICY_PYRAMID_IN_DETECT),
            "pyramid_in_detect_note": "-1 = the library's default (include/cusift_amd.h): a pipelining caller's call of >= 2 "
                                      "million pixels searches its octaves finest first and every detection launch also "
                                      "writes the next octave's image (ScaleDown's arithmetic, bit for bit) -- no "
                                      "ScaleDown launch, no memset",
        },
        # the contract's W + K steps with nothing before them but the set-up check (the like-for-like figure of rounds 1-4)
        "value_no_preflight_mpix_per_s": round(R.total_pix / (elapsed_literal / K) / 1e6, 2),
        "ms_per_step_no_preflight": round(elapsed_literal / K * 1e3, 4),
        "keypoints_per_s_in_hbm": round(total_kp / (elapsed / K), 1),
        "keypoints_per_step": total_kp,
    }
    # (N = 1: the prediction is for the wire format an N > 1 run of this command line would use)
    rec_b = gatherer.record_bytes if gatherer is not None else (
        160 if args.gather_compact else (588 if args.gather_exact else 540))
    out["gather_model"] = gather_model(local_kp, rec_b, extraction_only_ms if use_dist else R.ms_per_step,
                                       R.LAG if use_dist else E)
    if use_dist:
        out["gather_model"]["extraction_ms_per_step_source"] = (
            "the same K steps run without the exchange right behind the timed region (max over ranks)")
    out["gather_model"]["step_of_this_run_includes_an_exchange"] = bool(use_dist)
    if use_dist:
        out["config"]["gather_impl"] = R.gather_impl
        out["config"]["rccl_library"] = capi.Comm.library()
        out["config"]["gather_region_records"] = R.region_cap
        out["config"]["gather_record_bytes"] = rec_b
        out["config"]["gather_wire_format"] = (
            "%s%s" % (gatherer.wire_format, ", expanded on arrival to 588-byte SiftPoint records (extraction's fields; "
                      "the match fields arrive zeroed)" if gatherer.expand
                      else "")) if gatherer is not None else "exact (torch.distributed fallback)"
        # what the LIBRARY reports (ncclCommCount / ncclGetVersion), not this script's own bookkeeping: "RCCL saw N
        # ranks" can be read off the line
        info = comm.info() if comm is not None else {}
        out["config"]["rccl_ranks"] = info.get("lib_ranks")
        out["config"]["rccl_version"] = info.get("lib_version")
        out["config"]["ms_per_step_by_rank"] = [round(float(t) / K * 1e3, 4) for t in per_rank_elapsed]
        ex_ms = out["gather_model"]["ranks"].get(str(world), {}).get("eff_1.0", {}).get("ms_per_step_overlapped")
        if ex_ms:
            out["gather_model"]["measured_minus_predicted_ms_at_link_peak"] = round(R.ms_per_step - ex_ms, 4)
        if R.gather_waits is not None:
            out["config"]["gather_host_waits"] = R.gather_waits[0]
            out["config"]["gather_host_wait_ms"] = round(R.gather_waits[1], 3)


def teardown(R):
    dist = R.dist
    if R.world > 1:
        dist.barrier()  # the other ranks wait here while rank 0 runs its legs: communicators are torn down together
    for x in R.exs:
        x.close()
    if R.comm is not None:
        R.comm.close()
    if R.side_ctx is not None:
        R.side_ctx.close()
    if R.world > 1:
        dist.barrier()
        dist.destroy_process_group()
