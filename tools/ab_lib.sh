#!/bin/bash
# A/B of library builds on ONE box (device-to-device spread exceeds most effects): tools/ab_lib.sh libA.so libB.so ...
# prints the timed region and the single-stream stage table for each, interleaved, REPS times
for rep in $(seq 1 ${REPS:-3}); do for lib in "$@"; do
  CUSIFT_AMD_LIB=$PWD/$lib python bench.py --legs single --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('%-28s step %.4f single %.4f  down %.4f detect %.4f describe %.4f' % ('$lib', d['ms_per_step'], d['single_stream_leg']['ms_per_step'], s['scale_down'], s['detect_multi'], s['describe_all']))"
done; done
