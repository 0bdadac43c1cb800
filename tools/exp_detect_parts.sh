#!/bin/bash
# Upper bounds for detect_fused_kernel: builds without the refinement (1), without refinement + extrema analysis (2),
# with a window fill that hits the cache (3: VARIANTS=3),
# timed against the product build on one box (results wrong on purpose):  tools/exp_detect_parts.sh
set -u
SRC=cusift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fno-gpu-rdc"
/opt/rocm/bin/hipcc $FLAGS -o /tmp/libdet0.so $SRC/sift_capi.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip || exit 1
for v in ${VARIANTS:-1 2}; do
  /opt/rocm/bin/hipcc $FLAGS -DCUSIFT_DET_EXP=$v -o /tmp/libdet$v.so $SRC/sift_capi.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip \
      $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip || exit 1
done
for rep in 1 2 3; do for v in 0 ${VARIANTS:-1 2}; do
  CUSIFT_AMD_LIB=/tmp/libdet$v.so python bench.py --legs single --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('det_exp=$v detect', d['stage_ms_per_step']['detect_multi'], 'step', d['ms_per_step'], 'kp', d['keypoints_per_step'])"
done; done
