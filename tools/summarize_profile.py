#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh run (gpurun_out/<tag>/) into a small text summary for profiles/.

    python tools/summarize_profile.py gpurun_out/r01a profiles/r01a_summary.txt

Per kernel: calls, avg/min/max duration (kernel trace), and PMC counters averaged per launch.
For the stencil kernels the largest launches (octave 0 of the batch) are also listed on their own,
with HBM traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes; FETCH_SIZE reads half of a wide
coalesced stream on gfx950, MI355X_MICROARCH.md "HBM").
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    n = n.replace("cusift::", "").replace("void ", "")
    for k in ("detect_fused_kernel", "laplace_multi_fast_kernel"):  # template instances are one kernel here
        if n.startswith(k):
            n = k
    return n[:60]


def read_csv(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = []
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        cmd = "python3 bench.py --steps 5 --warmup 2 --legs single,two_stage"
        try:
            txt = open(os.path.join(src, "command.txt")).read()
            tr = [l for l in txt.splitlines() if l.startswith("trace args:")]
            cmd = "python3 bench.py " + (tr[0] if tr else txt).split(":", 1)[1].strip().splitlines()[0]
        except Exception:
            pass
        # per-kernel averages of the trace pass, for bench.py (roofline.profile_avg_launch_us) -> profiles/kernel_trace.json
        agg = defaultdict(lambda: [0, 0.0])
        for r in read_csv(stats[0]):
            a = agg[short(r["Name"])]
            a[0] += int(r["Calls"])
            a[1] += int(r["Calls"]) * float(r["AverageNs"]) / 1e3
        kt = {k: {"calls": c, "avg_us": t / c, "command": cmd} for k, (c, t) in agg.items() if c and k.endswith("_kernel")}
        try:  # the HIP-event figure of the SAME (profiled) process, for the record
            line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{") and '"metric"' in l][-1]
            rf = json.loads(line).get("roofline", {})
            if "avg_launch_ms" in rf and "laplace_multi_fast_kernel" in kt:
                kt["laplace_multi_fast_kernel"]["hip_event_avg_us_same_process"] = rf["avg_launch_ms"] * 1e3
        except Exception:  # noqa: BLE001
            pass
        try:  # ... and of the UN-profiled run of the same command on the same box (tools/profile_gpu.sh runs it first)
            rf = json.load(open(os.path.join(src, "bench.json"))).get("roofline", {})
            if "avg_launch_ms" in rf and "laplace_multi_fast_kernel" in kt:
                kt["laplace_multi_fast_kernel"]["hip_event_avg_us_same_box_unprofiled"] = rf["avg_launch_ms"] * 1e3
        except Exception:  # noqa: BLE001
            pass
        with open(os.path.join(os.path.dirname(dst) or ".", "kernel_trace.json"), "w") as f:
            json.dump(kt, f, indent=1)
        lines.append("== rocprofv3 --kernel-trace --stats (%s) ==" % cmd)
        lines.append("%-62s %6s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for r in read_csv(stats[0]):
            lines.append("%-62s %6s %12.2f %12.2f %12.2f %7s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                               r["Percentage"][:6]))
    single = glob.glob(os.path.join(src, "trace_single", "**", "*_kernel_stats.csv"), recursive=True)
    if single:
        lines.append("")
        lines.append("== rocprofv3 --kernel-trace --stats, ONE stream only (python3 bench.py --steps 10 --warmup 3 --streams 1 "
                     "--legs single --profile-run): no launch overlaps another, the averages are bench.py's HIP-event figures ==")
        lines.append("%-62s %6s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
        for r in read_csv(single[0]):
            lines.append("%-62s %6s %12.2f %12.2f %12.2f %7s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                               r["Percentage"][:6]))
        # the HIP-event figures bench.py measured in that same (profiled) process, beside the trace's averages
        try:
            line = [l for l in open(os.path.join(src, "trace_single.log")) if l.startswith("{") and '"metric"' in l][-1]
            b = json.loads(line)
            st = b["stage_ms_per_step"]
            by = defaultdict(list)
            for r in read_csv(single[0]):
                by[short(r["Name"])].append((int(r["Calls"]), float(r["AverageNs"]) / 1e3))
            lines.append("bench.py's HIP-event stage table in the same process (per step, `single` leg: one launch per octave, "
                         "stage timers on): scale_down %.1f us, detect %.1f us, describe_all %.1f us"
                         % (st["scale_down"] * 1e3, st["detect_multi"] * 1e3, st["describe_all"] * 1e3))
            da = by.get("describe_all_kernel")
            if da:
                lines.append("  describe_all_kernel: trace average %.1f us per launch (HIP events bracket the launch: + dispatch)"
                             % (sum(c * a for c, a in da) / sum(c for c, a in da)))
            df = by.get("detect_fused_kernel")
            if df and da:
                # one row per template instantiation: since round 5 <ident, 64, down> (octave 0), <general, 64, down>
                # (octaves 1 .. n-2) and <general, 64, no down> (the last octave)
                steps = sum(c for c, a in da)
                parts = sorted(df, key=lambda t: -t[1])
                lines.append("  detect_fused_kernel: %s = %.1f us per step in the trace (the stage table adds one dispatch per launch)"
                             % (" + ".join("%d x %.1f us" % (c // steps if c % steps == 0 else c, a) for c, a in parts),
                                sum(c * a for c, a in parts) / steps))
        except Exception as e:  # noqa: BLE001
            lines.append("(no bench line found in trace_single.log: %s)" % e)
    # PMC passes
    per_kernel = defaultdict(lambda: defaultdict(list))  # kernel -> counter -> [values per dispatch]
    per_dispatch = defaultdict(dict)  # (pass, dispatch) -> info
    single_valu = defaultdict(lambda: defaultdict(list))  # the one-stream PMC pass, kept apart: kernel -> counter -> values
    for f in glob.glob(os.path.join(src, "pmc_single", "**", "*_counter_collection.csv"), recursive=True):
        for r in read_csv(f):
            single_valu[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d) or os.path.basename(d) == "pmc_single":
            continue
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in read_csv(f):
                k = short(r["Kernel_Name"])
                per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (os.path.basename(d), r["Dispatch_Id"])
                per_dispatch[key]["kernel"] = k
                per_dispatch[key]["grid"] = int(r["Grid_Size"])
                per_dispatch[key]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                per_dispatch[key][r["Counter_Name"]] = float(r["Counter_Value"])
                per_dispatch[key]["vgpr"] = r.get("VGPR_Count")
                per_dispatch[key]["sgpr"] = r.get("SGPR_Count")
    ours = [k for k in per_kernel if k.endswith("_kernel") and "at::" not in k]
    lines.append("")
    lines.append("== PMC counters, mean per launch (separate --pmc passes) ==")
    for k in sorted(ours):
        lines.append(k)
        for c in sorted(per_kernel[k]):
            v = per_kernel[k][c]
            lines.append("    %-26s n=%-4d mean=%.4g  max=%.4g" % (c, len(v), sum(v) / len(v), max(v)))
    out_json = {}
    for kern in ("laplace_multi_fast_kernel", "find_points_fast_kernel", "detect_fused_kernel", "scale_down_kernel", "descriptors_kernel",
                 "orientations_kernel"):
        biggest = {}
        for (pas, disp), info in per_dispatch.items():
            if info.get("kernel") != kern:
                continue
            biggest[pas] = max(biggest.get(pas, 0), info["grid"])
        rows = defaultdict(list)
        for (pas, disp), info in per_dispatch.items():
            if info.get("kernel") == kern and info["grid"] == biggest.get(pas):
                for c, v in info.items():
                    if isinstance(v, float) and c != "dur_us":
                        rows[c].append(v)
                rows["dur_us(profiled)"].append(info["dur_us"])
                rows["_vgpr"].append(float(info["vgpr"] or 0))
                rows["_sgpr"].append(float(info["sgpr"] or 0))
        if not rows:
            continue
        lines.append("")
        lines.append("== %s: largest-grid launches only (octave 0 of the 64-image batch) ==" % kern)
        mean = {c: sum(v) / len(v) for c, v in rows.items()}
        for c in sorted(mean):
            lines.append("    %-26s %.6g" % (c, mean[c]))
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            fetch_b = mean["FETCH_SIZE"] * 1024
            write_b = mean["WRITE_SIZE"] * 1024
            lines.append("    HBM traffic per launch: read 2*FETCH_SIZE = %.1f MB, written WRITE_SIZE = %.1f MB, total %.1f MB"
                         % (2 * fetch_b / 1e6, write_b / 1e6, (2 * fetch_b + write_b) / 1e6))
            out_json[kern] = {"fetch_size_kib": mean["FETCH_SIZE"], "write_size_kib": mean["WRITE_SIZE"],
                              "hbm_bytes_octave0_launch": 2 * fetch_b + write_b}
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for (pas, disp), info in per_dispatch.items():
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in info:
                tot[info["kernel"]][c] += info[c]
                cnt[info["kernel"]][c] += 1
    lines.append("")
    for kern in ("laplace_multi_fast_kernel", "find_points_fast_kernel", "detect_fused_kernel"):
        if cnt[kern]["FETCH_SIZE"] and cnt[kern]["WRITE_SIZE"]:
            n = cnt[kern]["FETCH_SIZE"]
            per_launch = (2 * tot[kern]["FETCH_SIZE"] / n + tot[kern]["WRITE_SIZE"] / cnt[kern]["WRITE_SIZE"]) * 1024
            out_json.setdefault(kern, {})["hbm_bytes_per_launch"] = per_launch
            out_json[kern]["launches_profiled"] = n
            lines.append("%s: mean HBM traffic per launch over all %d profiled launches (all octaves) = %.1f MB"
                         % (kern, n, per_launch / 1e6))
    # VALU issue: wave-instructions per launch, per kernel (all launches).  (Rounds 1-2 also derived a "busy fraction"
    # SQ_ACTIVE_INST_VALU * 4 / SIMD-cycles: it exceeds 1 for the keypoint kernel -- not an occupancy of anything -- and is
    # gone; bench.py prints a mix-weighted issue bound instead, profiles/isa_mix.json.)
    valu_json = {"_source": "rocprofv3 --pmc SQ_INSTS_VALU ... (separate passes, --kernel-trace only) of `python3 bench.py "
                            "--steps 5 --warmup 2 --legs single,two_stage --profile-run`; see tools/profile_gpu.sh"}
    lines.append("")
    for kern in ("detect_fused_kernel", "describe_all_kernel", "laplace_multi_fast_kernel", "find_points_fast_kernel",
                 "scale_down_fast_kernel"):
        pk = per_kernel.get(kern)
        if not pk or "SQ_INSTS_VALU" not in pk:
            continue
        insts = sum(pk["SQ_INSTS_VALU"]) / len(pk["SQ_INSTS_VALU"])
        entry = {"valu_wave_insts_per_launch": insts, "launches_profiled": len(pk["SQ_INSTS_VALU"])}
        sv = single_valu.get(kern)
        if sv and "SQ_INSTS_VALU" in sv:
            # bench.py prices the launches of its ONE-STREAM leg: take the counts of the one-stream PMC pass (same launch
            # sequence, same chunk heights); the all-pass average above mixes in the four-stream region's taller chunks
            entry["valu_wave_insts_per_launch_all_passes"] = insts
            insts = sum(sv["SQ_INSTS_VALU"]) / len(sv["SQ_INSTS_VALU"])
            entry["valu_wave_insts_per_launch"] = insts
            entry["launches_profiled"] = len(sv["SQ_INSTS_VALU"])
            entry["counted_in"] = "the one-stream pass (--streams 1 --pyramid-in-detect 2 --legs single)"
        if "SQ_LDS_IDX_ACTIVE" in pk and "GRBM_GUI_ACTIVE" in pk:  # only when tools/pmc_lds.sh passes are present
            gui = sum(pk["GRBM_GUI_ACTIVE"]) / len(pk["GRBM_GUI_ACTIVE"])
            entry["lds_busy"] = round(sum(pk["SQ_LDS_IDX_ACTIVE"]) / len(pk["SQ_LDS_IDX_ACTIVE"]) / (256.0 * gui / 8.0), 4)
        if "SQ_WAVES" in pk:
            entry["waves_per_launch"] = sum(pk["SQ_WAVES"]) / len(pk["SQ_WAVES"])
        valu_json[kern] = entry
        lines.append("%s: %.4g VALU wave-instructions per launch (mean of %d launches)"
                     % (kern, insts, entry["launches_profiled"]))
    with open(os.path.join(os.path.dirname(dst) or ".", "valu.json"), "w") as f:
        json.dump(valu_json, f, indent=1)
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst, "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.splitext(dst)[0] + ".json", "w") as f:
        json.dump(out_json, f, indent=1)
    # what bench.py reads for `roofline.traffic` / `roofline_kernels[].hbm_traffic_bytes_per_launch`
    traffic = {"_source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `%s`; "
                          "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 averaged over all launches of the kernel "
                          "(all octaves); see %s and tools/profile_gpu.sh" % ("bench.py --steps 5 --warmup 2 --legs single,two_stage", os.path.basename(dst))}
    for kern, v in out_json.items():
        if "hbm_bytes_per_launch" in v:
            traffic[kern] = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "launches_profiled": v["launches_profiled"]}
    with open(os.path.join(os.path.dirname(dst) or ".", "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
