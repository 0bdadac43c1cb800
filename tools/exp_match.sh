#!/bin/bash
# Cost attribution of match_kernel: builds variants (-DCUSIFT_MATCH_EXP=n, results wrong on purpose) into /tmp and
# times tools/bench_match.py with each; run on the GPU box:  tools/exp_match.sh [sizes]
#   1 = no top-2 epilogue   2 = no barriers   3 = no MFMA (two packed multiplies per fragment instead)
set -u
SRC=cusift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fno-gpu-rdc"
for v in ${VARIANTS:-0 1 2 3}; do
  /opt/rocm/bin/hipcc $FLAGS -DCUSIFT_MATCH_EXP=$v -o /tmp/libmexp$v.so $SRC/sift_capi.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip \
      $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip || exit 1
done
for rep in 1 2; do for v in ${VARIANTS:-0 1 2 3}; do
  echo "exp=$v"; CUSIFT_AMD_LIB=/tmp/libmexp$v.so python tools/bench_match.py ${@:-16384}
done; done
