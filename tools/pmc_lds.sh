#!/bin/bash
# LDS-side counters of the keypoint kernel (separate --pmc passes, --kernel-trace only)
OUT=$PWD/gpurun_out/${1:-pmc_lds}; mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 bench.py --steps 3 --warmup 1 --legs single > $OUT/$tag.log 2>&1 || echo "pass failed: $set"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("cusift::","").replace("void ","")
        if "describe_all" in k or "detect_fused" in k:
            acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c,v in sorted(acc[k].items()): print("   %-28s n=%d mean=%.5g" % (c,len(v),sum(v)/len(v)))
PY
