#!/bin/bash
# parity of the keypoint kernels on the current build, then a same-box A/B of the libraries given as arguments
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
REPS=${REPS:-3} bash tools/ab_lib.sh "$@"
