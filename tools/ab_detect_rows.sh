#!/bin/bash
# needs the LAB build: python -m cusift_amd.build --lab && export CUSIFT_AMD_LIB=$PWD/cusift_amd/libcusift_amd_lab.so (the product library reads no tuning knob)
# chunk height of detect_fused_kernel: r = coef * sqrt(rows*strips*images), clamped to [2, HI]; timed region (2 streams) + single stream
for rep in 1 2; do for cfg in ${CFGS:-"0.022 24" "0.045 48" "0.06 64" "0.08 96" "0.1 128" "0.15 192"}; do
  set -- $cfg
  CUSIFT_DETECT_ROWS_COEF=$1 CUSIFT_DETECT_ROWS_HI=$2 python bench.py --legs single --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('coef=$1 hi=$2 ms_per_step', d['ms_per_step'], 'single', d['single_stream_leg']['ms_per_step'], 'detect', d['stage_ms_per_step']['detect_multi'])"
done; done
