#!/usr/bin/env python3
"""Kernel timeline of ONE 1080p extraction (BASELINE configs[1]) from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace1 -- python3 tools/trace_single_frame.py run
    python3 tools/trace_single_frame.py report gpurun_out/trace1

`run` extracts the same frame 30 times back to back with a wait after each; `report` prints the last dispatches of the run
with their durations and the gaps between them -- how many dispatches a frame is and what each costs."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    from cusift_amd import capi, synth
    w, h = 1920, 1080
    prm = capi.default_params(num_octaves=5, init_blur=1.0, peak_thresh=3.0, edge_thresh=10.0, max_pts=32768)
    p = capi.ialign_up(w, 128)
    src = np.zeros((h, p), dtype=np.float32)
    src[:, :w] = synth.tile(1000, w, h, preblur=1.0)
    with capi.Context(0) as ctx:
        d_img = capi.DeviceBuffer.from_numpy(ctx, src)
        d_pts = capi.DeviceBuffer(ctx, prm.max_pts * capi.SIFT_POINT_BYTES)
        d_cnt = capi.DeviceBuffer(ctx, 4)
        for _ in range(30):
            ctx.extract_batch(d_img.ptr, 1, w, h, p, h * p, prm, d_pts.ptr, d_cnt.ptr)
            ctx.synchronize()


def report(d, n_last=24):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n_last:]
    print("%-46s %9s %9s" % ("dispatch", "dur us", "gap us"))
    prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("cusift::", "")[:44]
        print("%-46s %9.1f %9.1f" % (name, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
        prev = e
    print("(the last %d dispatches of the run; a frame is the stretch between two long gaps -- the host's wait.  Under the "
          "profiler dispatches are serialised: durations include the dispatch itself)" % len(rows))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
