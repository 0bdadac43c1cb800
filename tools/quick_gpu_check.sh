#!/bin/bash
# parity tests that touch the keypoint kernels + the single-stream stage table; run on the GPU box
mkdir -p gpurun_out/quick
python -m pytest tests/test_gpu_parity.py tests/test_full_size_gpu.py tests/test_tiling_gpu.py -m gpu -q -x 2>&1 | tail -4
for rep in 1 2; do python bench.py --legs single --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'single', d['single_stream_leg']['ms_per_step'], d['stage_ms_per_step'])"; done
