#!/bin/bash
# needs the LAB build: python -m cusift_amd.build --lab && export CUSIFT_AMD_LIB=$PWD/cusift_amd/libcusift_amd_lab.so (the product library reads no tuning knob)
# chunk-height sweep of laplace_multi_fast_kernel with a given store policy, on ONE box
for aux in ${AUXES:-2 0}; do for r in 3 4 6 8 12 16 32; do
  CUSIFT_LAPLACE_AUX=$aux CUSIFT_LAPLACE_ROWS_LO=$r CUSIFT_LAPLACE_ROWS_HI=$r python bench.py --legs two_stage --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('aux=$aux rows=$r', r['achieved'], r['frac'], r['avg_launch_ms'], d['two_stage_leg']['find_points_GBps'])"
done; done
