#!/usr/bin/env python3
"""How busy the device is in bench.py's timed region, from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 30 --warmup 5 --legs none --profile-run
    python tools/timeline_busy.py gpurun_out/tl/*/*_kernel_trace.csv

Window: from the end of the 10th describe_all_kernel launch to the end of the last one.  Prints ms per step, the share
of the window in which at least one kernel runs, the share by number of kernels running at once, and every kernel's
summed span per step (spans stretch when kernels share the device: the sum exceeds the step)."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                 r["Kernel_Name"].split("(")[0].replace("cusift::", "").replace("void ", "")[:34]) for r in rows)
    desc = [e for e in ev if e[2].startswith("describe_all")]
    t0, t1 = desc[9][1], desc[-1][1]
    sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    steps = sum(1 for e in sel if e[2].startswith("describe_all"))
    busy, cs, ce = 0, None, None
    for s, e, _ in sel:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("%d steps, %.4f ms per step; at least one kernel running %.1f %% of the window" %
          (steps, (t1 - t0) / 1e6 / steps, 100.0 * busy / (t1 - t0)))
    pts = sorted([(s, 1) for s, _, _ in sel] + [(e, -1) for _, e, _ in sel])
    k, last, hist = 0, pts[0][0], collections.Counter()
    for t, d in pts:
        hist[k] += t - last
        last, k = t, k + d
    tt = sum(hist.values())
    print("share of the window by kernels running at once:", {k: round(v / tt, 3) for k, v in sorted(hist.items())})
    tot, cnt = collections.Counter(), collections.Counter()
    for s, e, n in sel:
        tot[n] += e - s
        cnt[n] += 1
    for n, v in tot.most_common(6):
        print("  %-34s %4d launches, %.3f ms of span per step, %.1f us each" % (n, cnt[n], v / 1e6 / steps, v / cnt[n] / 1e3))


if __name__ == "__main__":
    main()
