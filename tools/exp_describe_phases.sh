#!/bin/bash
# Cost attribution of describe_all_kernel: builds variants with one phase removed (-DCUSIFT_EXP=n, results wrong on
# purpose) into /tmp and times the single-stream leg with each; run on the GPU box:  tools/exp_describe_phases.sh
set -u
SRC=cusift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fno-gpu-rdc"
for v in 0 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc $FLAGS -DCUSIFT_EXP=$v -o /tmp/libexp$v.so $SRC/sift_capi.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip \
      $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip || exit 1
done
for rep in 1 2; do for v in 0 1 2 3 4 5 6; do
  CUSIFT_AMD_LIB=/tmp/libexp$v.so python bench.py --legs single --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('exp=$v describe_all', d['stage_ms_per_step']['describe_all'], 'detect', d['stage_ms_per_step']['detect_multi'], 'kp', d['keypoints_per_step'])"
done; done
