for rep in 1 2; do for hi in 8 10 12 16 24; do
# needs the LAB build: python -m cusift_amd.build --lab && export CUSIFT_AMD_LIB=$PWD/cusift_amd/libcusift_amd_lab.so (the product library reads no tuning knob)
  CUSIFT_LAPLACE_ROWS_HI=$hi python bench.py --legs two_stage --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rows_hi=$hi', r['achieved'], r['frac'], r['avg_launch_ms'], 'oct0', r['octave0_launch']['frac'], r['octave0_launch']['avg_launch_ms'])"
done; done
