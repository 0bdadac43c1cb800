#!/bin/bash
# A/B of variant builds under tools/ab/*.so on ONE box: parity smoke for each (the fixture + one 1080p image against the
# oracle), then tools/ab_lib.sh.   tools/ab_variants.sh base w1 w2 ...   -> gpurun_out/ab/<tag>.txt
set -u
tag=${AB_TAG:-ab}
mkdir -p gpurun_out/ab
out=gpurun_out/ab/$tag.txt
: > $out
libs=""
for v in "$@"; do
  libs="$libs tools/ab/$v.so"
  if [ "${AB_PARITY:-1}" = "1" ]; then
    CUSIFT_AMD_LIB=$PWD/tools/ab/$v.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
      -k "extract_fixture_matches_oracle or extract_1080p_matches_oracle or two_stage_and_fused or root_sift" 2>&1 | tail -2 | sed "s/^/$v parity: /" >> $out
  fi
done
REPS=${REPS:-3} bash tools/ab_lib.sh $libs >> $out 2>&1
cat $out
