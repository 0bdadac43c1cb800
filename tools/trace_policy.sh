#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of tools/ab_pyramid.py for one policy at a time (single stream), so
# that the per-kernel averages of the ScaleDown-chain-first sequence and of the pyramid-in-detection sequence sit side by side.
# Usage: tools/trace_policy.sh <outdir under gpurun_out> [content]
set -u
OUT=$PWD/gpurun_out/${1:-r05/trace_policy}
CONTENT=${2:-tile}
mkdir -p "$OUT"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8 HSA_ENABLE_IPC_MODE_LEGACY=0
for P in 0 2; do
  AB_POLICIES=$P AB_STREAMS=${AB_STREAMS:-1} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p$P" -- python3 tools/ab_pyramid.py 1 "$CONTENT" > "$OUT/p$P.log" 2>&1 </dev/null
  f=$(ls "$OUT"/p$P/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== policy $P ($CONTENT)"
  if [ -n "$f" ]; then cp "$f" "$OUT/p${P}_kernel_stats.csv"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %6s  avg %10.1f us  total %8.2f ms  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
  else echo "no stats file"; tail -5 "$OUT/p$P.log"; fi
  rm -rf "$OUT/p$P"
done
