#!/bin/bash
# What does describe_all_kernel's time depend on?  Builds with 200 extra independent v_fma_f32 per keypoint (+12 % of
# its vector instructions) placed in the sampling phase (7), in the gather (8) or after the normalisation (9), timed
# against the product build (0) on one box:  tools/exp_describe_ballast.sh
set -u
SRC=cusift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fno-gpu-rdc"
for v in 0 7 8 9; do
  /opt/rocm/bin/hipcc $FLAGS -DCUSIFT_EXP=$v -o /tmp/libexp$v.so $SRC/sift_capi.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip \
      $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip || exit 1
done
for rep in 1 2 3; do for v in 0 7 8 9; do
  CUSIFT_AMD_LIB=/tmp/libexp$v.so python bench.py --legs single --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('exp=$v describe_all', d['stage_ms_per_step']['describe_all'], 'step', d['ms_per_step'])"
done; done
