"""One stream, HIP events per launch: the per-stage table of the timed step, what a lone caller gets, and the VALU
rooflines of the two kernels that own the step (instruction counts from the committed PMC passes, times from this run)."""
import time

from .common import FP32_VALU_PEAK_TF, PRETEST_SKIP_HEADLINE


def run(R):
    args, torch, capi, out, ex, d_imgs, K = R.args, R.torch, R.capi, R.out, R.ex, R.d_imgs, R.args.steps
    B, w, h, E, local_kp = R.B, R.w, R.h, R.E, R.local_kp
    valu, traffic, isa_mix, down_b = R.valu, R.traffic, R.isa_mix, R.down_b
    run_single_stream, stage_table = R.run_single_stream, R.stage_table
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
    R.stage = stage
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
