#!/bin/bash
# A/B of compiler scheduling options on one box: builds the library with each extra flag set into /tmp and times the
# single-stream stage table (tools/ab_flags.sh; results identical by construction -- only instruction order changes)
set -u
SRC=cusift_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -fno-gpu-rdc"
i=0
declare -a NAMES
while IFS= read -r extra; do
  /opt/rocm/bin/hipcc $FLAGS $extra -o /tmp/libflag$i.so $SRC/sift_context.hip $SRC/sift_stages.hip $SRC/sift_driver.hip $SRC/sift_stencils.hip $SRC/sift_keypoints.hip \
      $SRC/sift_match.hip $SRC/sift_frontend.hip $SRC/sift_homography.hip $SRC/sift_comm.hip 2>/dev/null || { echo "build failed: $extra"; continue; }
  NAMES[$i]="$extra"; i=$((i+1))
done <<'LIST'

-mllvm -amdgpu-sched-strategy=max-ilp
-mllvm -amdgpu-schedule-metric-bias=0
-mllvm -amdgpu-sched-strategy=max-memory-clause
-mllvm -amdgpu-schedule-relaxed-occupancy
LIST
for rep in 1 2; do for k in $(seq 0 $((i-1))); do
  CUSIFT_AMD_LIB=/tmp/libflag$k.so python bench.py --legs single --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('%-52s step %.4f  detect %.4f describe %.4f kp %d' % ('[${NAMES[$k]}]', d['ms_per_step'], s['detect_multi'], s['describe_all'], d['keypoints_per_step']))"
done; done
