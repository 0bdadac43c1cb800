#!/bin/bash
# builds libcusift_amd with -DCUSIFT_EXP=10 into /tmp and prints the per-segment cycle table (GPU box)
set -u
SRC=cusift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fno-gpu-rdc"
/opt/rocm/bin/hipcc $FLAGS -DCUSIFT_EXP=${1:-10} -o /tmp/libexp10.so $SRC/sift_capi.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip \
    $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip || exit 1
CUSIFT_AMD_LIB=/tmp/libexp10.so python tools/exp_describe_stamps.py
