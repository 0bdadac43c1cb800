#!/bin/bash
# builds libcusift_amd with -DCUSIFT_DET_STAMPS into /tmp and prints the per-segment cycle table (GPU box)
set -u
SRC=cusift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fno-gpu-rdc"
/opt/rocm/bin/hipcc $FLAGS -DCUSIFT_DET_STAMPS=${1:-1} -o /tmp/libdet.so $SRC/sift_capi.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip \
    $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip || exit 1
DET_STAMPS_MODE=${1:-1} CUSIFT_AMD_LIB=/tmp/libdet.so python tools/exp_detect_stamps.py 1
