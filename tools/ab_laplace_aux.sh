#!/bin/bash
# needs the LAB build: python -m cusift_amd.build --lab && export CUSIFT_AMD_LIB=$PWD/cusift_amd/libcusift_amd_lab.so (the product library reads no tuning knob)
# A/B of the DoG stores' cache policy in laplace_multi_fast_kernel on ONE box (device-to-device spread is larger than the effect)
for rep in 1 2; do for aux in 0 2 16 18; do
  CUSIFT_LAPLACE_AUX=$aux python bench.py --legs two_stage --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('aux=$aux', r['achieved'], r['frac'], r['avg_launch_ms'], d['two_stage_leg']['find_points_GBps'])"
done; done
