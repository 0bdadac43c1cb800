#!/usr/bin/env python3
"""Regenerates the measured numbers in the documentation from the records under profiles/r06/ -- nothing in DESIGN.md or
README.md that was MEASURED is typed by hand.

    python tools/gen_numbers.py            # rewrites the blocks between <!-- numbers:begin --> / <!-- numbers:end --> in
                                           # DESIGN.md and between <!-- headline:begin --> / <!-- headline:end --> in README.md
    python tools/gen_numbers.py --check    # exit 1 if a block differs from what the records give (tests/test_docs_numbers.py)

Records (all written on an MI355X box by the commands named in profiles/README.md):
  bench_steps20_warmup5.json   python bench.py --steps 20 --warmup 5      (the driver's command)
  bench_default.json           python bench.py
  cpp_threads.txt              tests/cpp/threads_dropin + pipeline_dropin
  golden_shares.json           tools/golden_shares.py (CPU: the oracle against both golden files)
  gpu_tests_tail.txt, fuzz_parity.txt, fuzz_tiled.txt, fuzz_match.txt
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R06 = os.path.join(ROOT, "profiles", "r06")


def load(name):
    return json.load(open(os.path.join(R06, name)))


def last_line(name):
    try:
        lines = [l.strip() for l in open(os.path.join(R06, name)) if l.strip()]
        return lines[-1]
    except OSError:
        return "(no record)"


def g(d, *path, default=None):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return default
        d = d[p]
    return d


def f(x, nd=1):
    return "n/a" if x is None else ("%." + str(nd) + "f") % x


def gp(mpix):  # Mpix/s -> Gpix/s text
    return "n/a" if mpix is None else "%.1f" % (mpix / 1e3)


def numbers_block():
    drv = load("bench_steps20_warmup5.json")
    dfl = load("bench_default.json")
    L = []
    a = L.append
    a("Records: `profiles/r06/bench_steps20_warmup5.json` (the driver's command, `python bench.py --steps 20 --warmup 5`) and")
    a("`profiles/r06/bench_default.json` (`python bench.py`: 50 steps) — one MI355X each, the builder's boxes; rates differ by")
    a("3–7 % from box to box. Gpix/s = 1e9 base pixels per second; one step = 64 × 1920 × 1080 = 132.7 Mpix.")
    a("")
    a("| the step (64 × 1080p, SiftData left in HBM) | driver's form | default form |")
    a("|---|---|---|")
    a("| **`value`** (steady clocks, four streams) | **%s Gpix/s**, %s ms per step | %s Gpix/s, %s ms |"
      % (gp(drv["value"]), f(drv["ms_per_step"], 4), gp(dfl["value"]), f(dfl["ms_per_step"], 4)))
    a("| `value_no_preflight_mpix_per_s` (W + K steps as the contract words them) | %s Gpix/s, %s ms | %s Gpix/s, %s ms |"
      % (gp(drv.get("value_no_preflight_mpix_per_s")), f(drv.get("ms_per_step_no_preflight"), 4),
         gp(dfl.get("value_no_preflight_mpix_per_s")), f(dfl.get("ms_per_step_no_preflight"), 4)))
    a("| the same region started from an idle device | %s ms | %s ms |"
      % (f(drv.get("ms_per_step_from_idle"), 4), f(dfl.get("ms_per_step_from_idle"), 4)))
    sp = lambda d: g(d, "ms_per_step_spread") or {}
    a("| five regions: min / median / max ms per step | %s / %s / %s | %s / %s / %s |"
      % (f(sp(drv).get("min"), 4), f(sp(drv).get("median"), 4), f(sp(drv).get("max"), 4),
         f(sp(dfl).get("min"), 4), f(sp(dfl).get("median"), 4), f(sp(dfl).get("max"), 4)))
    a("| keypoints per step; keypoints/s in HBM | %s; %s M/s | %s; %s M/s |"
      % (drv.get("keypoints_per_step"), f(drv.get("keypoints_per_s_in_hbm", 0) / 1e6, 1),
         dfl.get("keypoints_per_step"), f(dfl.get("keypoints_per_s_in_hbm", 0) / 1e6, 1)))
    ss = lambda d: g(d, "single_stream_leg") or {}
    a("| a caller with ONE batch in flight (`lone_caller_ms_per_step`) | %s ms | %s ms |"
      % (f(ss(drv).get("lone_caller_ms_per_step"), 4), f(ss(dfl).get("lone_caller_ms_per_step"), 4)))
    bc = lambda d, k: g(d, k)
    a("| by content: `blobs` / raw tiles (every image saturates `maxPts`) / initBlur = 0 | %s / %s / %s Gpix/s | %s / %s / %s |"
      % (gp(bc(drv, "value_blobs_mpix_per_s")), gp(bc(drv, "value_tile_raw_mpix_per_s")), gp(bc(drv, "value_initblur0_mpix_per_s")),
         gp(bc(dfl, "value_blobs_mpix_per_s")), gp(bc(dfl, "value_tile_raw_mpix_per_s")), gp(bc(dfl, "value_initblur0_mpix_per_s"))))
    a("| keypoints/s host-visible: exact 588-B / trimmed 540-B records | %s / %s M/s | %s / %s M/s |"
      % (f((drv.get("keypoints_per_s_host_visible") or 0) / 1e6, 1), f((drv.get("keypoints_per_s_host_visible_trimmed") or 0) / 1e6, 1),
         f((dfl.get("keypoints_per_s_host_visible") or 0) / 1e6, 1), f((dfl.get("keypoints_per_s_host_visible_trimmed") or 0) / 1e6, 1)))
    hh = lambda d, k: g(d, "host_to_host", k) or {}
    a("| **PCIe-inclusive, host to host** (never `value`): 8-bit frames in pinned memory → SiftData in pinned memory | %s Gpix/s (%s of the upload alone); C ABI alone %s | %s (%s); %s |"
      % (gp(hh(drv, "u8").get("Mpix_per_s")), f(g(hh(drv, "u8"), "bound", "frac_of_bound"), 2), gp(hh(drv, "u8_c_abi").get("Mpix_per_s")),
         gp(hh(dfl, "u8").get("Mpix_per_s")), f(g(hh(dfl, "u8"), "bound", "frac_of_bound"), 2), gp(hh(dfl, "u8_c_abi").get("Mpix_per_s"))))
    a("| … float32 frames (the reference's entry point type) | %s Gpix/s | %s |"
      % (gp(hh(drv, "f32").get("Mpix_per_s")), gp(hh(dfl, "f32").get("Mpix_per_s"))))
    a("")
    r = drv.get("roofline", {})
    rd = dfl.get("roofline", {})
    a("**Roofline of the blur + DoG kernel** (`laplace_multi_fast_kernel`, HBM-bound; algorithmic %s B per launch):"
      % "{:,}".format(r.get("algorithmic_bytes_per_launch", 0)))
    a("driver's form **%s GB/s = %s of 8 TB/s** (HIP-event average %s µs per launch, %s launches); default form %s GB/s = %s."
      % (f(r.get("achieved"), 1), f(r.get("frac"), 3), f((r.get("avg_launch_ms") or 0) * 1e3, 1), r.get("launches"),
         f(rd.get("achieved"), 1), f(rd.get("frac"), 3)))
    a("Committed `rocprofv3 --kernel-trace --stats` pass of the same command (`profiles/r06_final_summary.txt`,")
    a("`profiles/kernel_trace.json`): %s µs per launch over %s launches; HIP events inside that profiled process read %s µs"
      % (f(r.get("profile_avg_launch_us"), 1), r.get("profile_launches"), f(r.get("profile_process_hip_event_avg_us"), 1)))
    try:
        pp = load("profile_vs_live_pairs.json")["pairs"]
        a("= %s × the trace — and %s in the five profile runs of the round (one dispatch per launch): the two ways of timing the"
          % (f(r.get("hip_events_over_trace_same_process"), 3),
             " / ".join("%.3f" % (q["live_avg_us_inside_the_profiled_process"] / q["trace_avg_us"]) for q in pp)))
        a("kernel agree to 0.1 %%. Between separate runs its average moves by ±5 %% (trace %s µs; un-profiled runs on the same boxes"
          % " / ".join("%.0f" % q["trace_avg_us"] for q in pp))
        a("%s µs; this line's %s µs): clocks, not method (`profiles/r06/profile_vs_live_pairs.json`). PMC traffic %s B per launch ="
          % (" / ".join("%.0f" % q["live_avg_us_unprofiled_same_box"] for q in pp), f((r.get("avg_launch_ms") or 0) * 1e3, 0),
             "{:,}".format(int(r.get("traffic") or 0))))
    except OSError:
        a("PMC traffic %s B per launch =" % "{:,}".format(int(r.get("traffic") or 0)))
    a("%s × algorithmic. Octave-0 launch alone: %s GB/s = %s."
      % (f((r.get("traffic") or 0) / max(1, r.get("algorithmic_bytes_per_launch", 1)), 2),
         f(g(r, "octave0_launch", "achieved"), 1), f(g(r, "octave0_launch", "frac"), 3)))
    a("")
    a("**The timed step's own kernels** (one stream, HIP events per launch; instruction counts from the committed PMC pass):")
    a("")
    a("| kernel | ms per step | of the fp32 vector peak (157.3 TFLOP/s) | of its mix-weighted issue bound |")
    a("|---|---|---|---|")
    for rk in drv.get("roofline_kernels", []):
        a("| `%s` | %s | %s | %s |" % (rk["kernel"], f(rk["ms_per_step"], 4), f(rk["frac"], 3),
                                         f(g(rk, "issue_bound", "frac_of_issue_bound"), 3)))
    kk = [rk for rk in drv.get("roofline_kernels", []) if "valu_wave_insts_per_keypoint" in rk]
    if kk:
        a("")
        a("`describe_all_kernel`: %s vector wave-instructions per keypoint (%s keypoints per step)."
          % (f(kk[0]["valu_wave_insts_per_keypoint"], 0), drv.get("keypoints_per_step")))
    a("")
    a("**BASELINE configs as driver-run legs** (`config_legs`, driver's form):")
    a("")
    c = drv.get("config_legs", {})
    c0, c1, c4 = c.get("configs[0]", {}), c.get("configs[1]", {}), c.get("configs[4]", {})
    cpu0 = g(drv, "cpu_baseline", "configs0") or {}
    a("| config | measured |")
    a("|---|---|")
    a("| [0] 640 × 480 fixture, 3 octaves, host image in → host SiftData out | HIP %s ms per image (%s keypoints); CPU oracle on one core %s ms (%s keypoints, counts equal: %s) |"
      % (f(c0.get("hip_ms_per_image_median"), 4), c0.get("keypoints"), f(cpu0.get("ms_per_image"), 1), cpu0.get("keypoints"),
         cpu0.get("keypoints_equal_hip")))
    a("| [1] ONE 1920 × 1080 frame, device-resident | latency %s ms median / %s p95; back to back %s ms per frame = %s Gpix/s; %s keypoints |"
      % (f(g(c1, "eager", "latency_ms_median"), 4), f(g(c1, "eager", "latency_ms_p95"), 4),
         f(g(c1, "eager", "back_to_back_ms_per_frame"), 4), gp(g(c1, "eager", "back_to_back_mpix_per_s")), c1.get("keypoints")))
    a("| [1] … host float image in → host SiftData out (`SiftData::Extract`'s shape) | %s ms pageable, %s ms into pinned records (%s / %s Gpix/s) |"
      % (f(g(c1, "host_float_in_host_siftdata_out", "latency_ms_median"), 4),
         f(g(c1, "host_float_in_pinned_host_siftdata_out", "latency_ms_median"), 4),
         gp(g(c1, "host_float_in_host_siftdata_out", "Mpix_per_s")), gp(g(c1, "host_float_in_pinned_host_siftdata_out", "Mpix_per_s"))))
    w = c4.get("whole_on_one_gpu", {})
    v = c4.get("virtual_ranks_on_one_gpu", {})
    a("| [2] 64 × 1080p on one GPU | the timed region above |")
    a("| [3] 512 × 1080p over 8 GPUs + all-gatherv | **not measured: no multi-GPU node**; `gather_model` below |")
    a("| [4] ONE 8192 × 8192 image, whole on one GPU | %s ms = %s Gpix/s, %s keypoints |"
      % (f(w.get("ms_per_image"), 4), gp(w.get("Mpix_per_s")), w.get("keypoints")))
    a("| [4] … 8 virtual ranks on one GPU (exchanges = device copies, run one after the other) | %s ms in all, ≈ %s per rank; keypoints equal the whole image: %s |"
      % (f(v.get("total_ms"), 3), f(v.get("per_rank_ms_estimate"), 3), v.get("keypoints_equal_whole_image")))
    a("| [4] … tiled over 8 GPUs | **not measured**; `tiled_model` below |")
    a("")
    tm = drv.get("tiled_model", {})
    t8 = g(tm, "ranks", "8") or {}
    a("`tiled_model` (8 ranks, from the measured one-GPU time): kernels %s ms per rank; halo exchanges + merge %s–%s ms;"
      % (f(t8.get("kernel_ms_per_rank"), 3),
         f(g(t8, "latency_20us_eff_1.0", "predicted_ms_per_image", default=0) - (t8.get("kernel_ms_per_rank") or 0), 2),
         f(g(t8, "latency_60us_eff_0.7", "predicted_ms_per_image", default=0) - (t8.get("kernel_ms_per_rank") or 0), 2)))
    a("predicted %s–%s ms per image = **%s×–%s× over one GPU**."
      % (f(g(t8, "latency_20us_eff_1.0", "predicted_ms_per_image"), 3), f(g(t8, "latency_60us_eff_0.7", "predicted_ms_per_image"), 3),
         f(g(t8, "latency_60us_eff_0.7", "predicted_speedup_over_one_gpu"), 2), f(g(t8, "latency_20us_eff_1.0", "predicted_speedup_over_one_gpu"), 2)))
    gm = drv.get("gather_model", {})
    g8 = g(gm, "ranks", "8") or {}
    a("`gather_model` (configs[3], %s-byte records, %s keypoints per rank and step): one shard over one link %s ms at link peak"
      % (gm.get("record_bytes"), gm.get("records_per_rank_per_step"), f(g(g8, "eff_1.0", "exchange_ms"), 3)))
    a("against %s ms of extraction — predicted weak-scaling efficiency at 8 ranks %s (link peak) / %s (70 %% RCCL efficiency)."
      % (f(gm.get("extraction_ms_per_step"), 3), f(g(g8, "eff_1.0", "weak_scaling_efficiency_overlapped"), 2),
         f(g(g8, "eff_0.7", "weak_scaling_efficiency_overlapped"), 2)))
    a("")
    cb = drv.get("cpu_baseline", {})
    a("**CPU baseline** (`kind: \"%s\"`: the oracle, not OpenCV — `opencv.available: %s`): %s Mpix/s on %s threads (%s usable CPUs)."
      % (cb.get("kind"), g(cb, "opencv", "available"), f(cb.get("value"), 1), cb.get("cores"), cb.get("usable_cpus")))
    a("")
    a("**An unchanged cuSIFT program from N host threads** (`tests/cpp/threads_dropin`, one 1080p frame per `ExtractSift` call,")
    a("device-resident `cuImage`, host records read back every call; `profiles/r06/cpp_threads.txt`):")
    a("")
    try:
        for line in open(os.path.join(R06, "cpp_threads.txt")):
            m = re.match(r"threads: (\d+) threads x (\d+) frames (\d+x\d+): one thread ([0-9.]+) ms per frame \(([0-9.]+) Gpix/s\), "
                         r"\d+ threads ([0-9.]+) ms per frame \(([0-9.]+) Gpix/s\), (\d+) keypoints, (.*)", line.strip())
            if m and int(m.group(1)) > 1:
                a("* %s threads: %s ms per frame = **%s Gpix/s** (one thread: %s ms = %s Gpix/s); results %s"
                  % (m.group(1), m.group(6), m.group(7), m.group(4), m.group(5), m.group(9)))
            m = re.match(r"pipeline: (\d+) contexts, .*: ([0-9.]+) ms per batch, ([0-9.]+) Gpix/s", line.strip())
            if m:
                a("* for scale, the batch entry point from C++ (`pipeline_dropin`, %s contexts, 64 frames per call): %s ms per batch = %s Gpix/s"
                  % (m.group(1), m.group(2), m.group(3)))
    except OSError:
        a("(no record)")
    a("")
    a("**Parity records of the final build.**")
    gs = None
    try:
        gs = load("golden_shares.json")
    except OSError:
        pass
    if gs:
        for name in ("cusift1_check", "cusift1"):
            s = gs[name]
            a("* the reference's `%s` (4,096 rows) vs the oracle = the HIP path (bit-identical in these fields): coarse rows %s %% within 1e-3 px;"
              " octave-0 rows %s %% found; orientation within 1e-3 degree %s %% (coarse) / %s %% (octave 0), within 1 degree %s %% / %s %%,"
              " median %s / %s degree; rows further than 1 degree: %s, of which at the oracle's second peak: %s"
              % (name, f(100 * s["coarse_within_1e-3"], 2), f(100 * s["oct0_found_1e-2"], 2), f(100 * s["coarse_ori_lt_1e-3"], 2),
                 f(100 * s["oct0_ori_lt_1e-3"], 2), f(100 * s["coarse_ori_lt_1"], 2), f(100 * s["oct0_ori_lt_1"], 2),
                 "%.1e" % s["coarse_ori_median"], "%.1e" % s["oct0_ori_median"], s["tail"]["outliers_gt_1_deg"],
                 s["tail"]["outliers_at_second_peak"]))
    a("* `pytest -m gpu`: %s" % last_line("gpu_tests_tail.txt"))
    a("* `tools/fuzz_parity.py` (1,500 + 2,500 + 4,000 cases, three seeds): %s" % last_line("fuzz_parity.txt"))
    a("* `tools/fuzz_tiled.py 40`: %s" % last_line("fuzz_tiled.txt"))
    a("* `tools/fuzz_match.py 100`: %s" % last_line("fuzz_match.txt"))
    return "\n".join(L)


def headline_block():
    drv = load("bench_steps20_warmup5.json")
    r = drv.get("roofline", {})
    c1 = g(drv, "config_legs", "configs[1]") or {}
    w = g(drv, "config_legs", "configs[4]", "whole_on_one_gpu") or {}
    L = ["(generated by `tools/gen_numbers.py` from `profiles/r06/bench_steps20_warmup5.json`, the driver's command on one of the",
         "builder's MI355X boxes; box-to-box spread 3–7 %; everything else: `DESIGN.md` §9)", "",
         "* 64 × 1080p per step, SiftData in HBM: **%s Gpix/s** (%s ms per step, %s M keypoints/s) at steady clocks; %s Gpix/s"
         % (gp(drv["value"]), f(drv["ms_per_step"], 4), f(drv.get("keypoints_per_s_in_hbm", 0) / 1e6, 0),
            gp(drv.get("value_no_preflight_mpix_per_s"))),
         "  for the literal `--warmup 5 --steps 20`; by content %s (raw tiles) … %s Gpix/s"
         % (gp((drv.get("value_range_mpix_per_s") or [None, None])[0]), gp((drv.get("value_range_mpix_per_s") or [None, None])[1])),
         "* blur + DoG kernel: **%s of the 8 TB/s HBM roofline** (%s GB/s algorithmic; PMC traffic %s × algorithmic)"
         % (f(r.get("frac"), 3), f(r.get("achieved"), 0), f((r.get("traffic") or 0) / max(1, r.get("algorithmic_bytes_per_launch", 1)), 2)),
         "* one 1080p frame: %s ms latency, %s Gpix/s back to back; one 8192² image: %s ms (%s Gpix/s)"
         % (f(g(c1, "eager", "latency_ms_median"), 3), gp(g(c1, "eager", "back_to_back_mpix_per_s")), f(w.get("ms_per_image"), 3),
            gp(w.get("Mpix_per_s"))),
         "* host to host over PCIe (8-bit frames in, SiftData out): %s Gpix/s"
         % gp(g(drv, "host_to_host", "u8", "Mpix_per_s")),
         "* CPU oracle beside it: %s Mpix/s on %s threads (not OpenCV: absent)"
         % (f(g(drv, "cpu_baseline", "value"), 0), g(drv, "cpu_baseline", "cores"))]
    return "\n".join(L)


def splice(path, tag, body, check):
    text = open(path).read()
    pat = re.compile(r"(<!-- %s:begin -->\n)(.*?)(\n<!-- %s:end -->)" % (tag, tag), re.S)
    m = pat.search(text)
    if not m:
        raise SystemExit("%s: no <!-- %s:begin/end --> block" % (path, tag))
    if check:
        return m.group(2) == body
    open(path, "w").write(text[:m.start(2)] + body + text[m.end(2):])
    return True


def main():
    check = "--check" in sys.argv
    ok = splice(os.path.join(ROOT, "DESIGN.md"), "numbers", numbers_block(), check)
    ok = splice(os.path.join(ROOT, "README.md"), "headline", headline_block(), check) and ok
    if check and not ok:
        print("documentation numbers are stale: run python tools/gen_numbers.py")
        raise SystemExit(1)


if __name__ == "__main__":
    main()
