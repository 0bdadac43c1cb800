#!/bin/bash
# VALU wave-instructions per describe_all / detect launch for several library builds (run on the GPU box):
#   tools/pmc_valu_ab.sh libA.so libB.so ...      (--pmc pass with --kernel-trace only)
set -u
cd /tmp 2>/dev/null && cd - >/dev/null
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  out=$PWD/gpurun_out/pmc_ab/$tag
  mkdir -p "$out"
  export CUSIFT_AMD_LIB=$PWD/$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d "$out" -- python3 bench.py --legs none --steps 3 --warmup 1 > "$out/log.txt" 2>&1
  python3 - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("cusift::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    if "describe_all" in k or "detect_fused" in k:
        print(tag, k, {c: "%.4g x%d" % (sum(x) / len(x), len(x)) for c, x in v.items()})
PY
done
