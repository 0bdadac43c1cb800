#!/bin/bash
# needs the LAB build: python -m cusift_amd.build --lab && export CUSIFT_AMD_LIB=$PWD/cusift_amd/libcusift_amd_lab.so (the product library reads no tuning knob)
# A/B of library builds on the two-stage leg (the blur+DoG roofline exhibit) on ONE box: tools/ab_two_stage.sh a.so b.so
for rep in $(seq 1 ${REPS:-3}); do for lib in "$@"; do
  CUSIFT_AMD_LIB=$PWD/$lib python bench.py --legs two_stage --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-22s' % '$lib', r['achieved'], r['frac'], r['avg_launch_ms'], 'oct0', r['octave0_launch']['frac'], r['octave0_launch']['avg_launch_ms'], 'find', d['two_stage_leg']['find_points_GBps'])"
done; done
