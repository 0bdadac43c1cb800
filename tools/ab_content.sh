#!/bin/bash
# A/B of library builds on ONE box over the three contents (headline tiles, blobs, raw tiles): timed region, the
# single-stream stage table and the content legs' pipelined rates.  tools/ab_content.sh libA.so libB.so ...
for rep in $(seq 1 ${REPS:-2}); do for lib in "$@"; do
  CUSIFT_AMD_LIB=$PWD/$lib python bench.py --legs single,content --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; c=d['content_legs']
b=[v for k,v in c.items() if k.startswith('blobs')][0]; r=[v for k,v in c.items() if k.startswith('tile_raw')][0]
print('%-22s step %.4f | detect %.4f describe %.4f | blobs pipelined %.4f (det %.4f desc %.4f) | raw pipelined %.4f (det %.4f desc %.4f)' % ('$lib', d['ms_per_step'], s['detect_multi'], s['describe_all'], b['ms_per_step_pipelined'], b['stage_ms_per_step']['detect_multi'], b['stage_ms_per_step']['describe_all'], r['ms_per_step_pipelined'], r['stage_ms_per_step']['detect_multi'], r['stage_ms_per_step']['describe_all']))"
done; done
