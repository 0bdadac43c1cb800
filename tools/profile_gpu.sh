#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the benchmark.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
# Results land under gpurun_out/<tag>/ ; tools/summarize_profile.py condenses them into profiles/.
# PMC passes use --kernel-trace only (never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp 2>/dev/null && cd - >/dev/null
export TMPDIR=/tmp
# The benchmark sets these with os.environ.setdefault() -- too late under `rocprofv3 --pmc`, whose preloaded library
# initialises the GPU before Python runs: until round 5 the PMC passes ran on 4 hardware queues and the un-profiled bench
# on 8.  Every pass below runs with 8 (profiles/README.md says so per pass).
export GPU_MAX_HW_QUEUES=8 HSA_ENABLE_IPC_MODE_LEGACY=0
echo "GPU_MAX_HW_QUEUES=$GPU_MAX_HW_QUEUES HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY (every pass)" > "$OUT/environment.txt"
BENCH_ARGS="--steps 5 --warmup 2 --legs single,two_stage --profile-run $*"
# the kernel-trace pass runs the DRIVER's steps (--steps 20 --warmup 5), behind the same set-up check and pre-flight as the
# un-profiled command: its per-kernel averages are at the clocks the line's numbers are measured at (round 5's pass ran 5
# steps from a cold start and read the blur + DoG kernel 13 % slower than the driver's line)
TRACE_ARGS="--steps 20 --warmup 5 --legs single,two_stage --profile-run $*"
echo "bench args: $BENCH_ARGS" > "$OUT/command.txt"
echo "trace args: $TRACE_ARGS" >> "$OUT/command.txt"
echo "progress: un-profiled bench" 
python3 bench.py $TRACE_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "progress: kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $TRACE_ARGS > "$OUT/trace.log" 2>&1
# ONE stream only (timed region and single-stream leg both on one stream): kernel spans do not overlap, so this table's
# averages are the HIP-event figures of bench.py's stage_ms_per_step (the pass above mixes overlapped and lone launches)
echo "progress: one-stream kernel trace"
# (--pyramid-in-detect 2: a lone caller's default would be "octave 0 only"; these two passes are about the timed region's kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_single" -- python3 bench.py --steps 10 --warmup 3 --streams 1 --pyramid-in-detect 2 --legs single --profile-run $* > "$OUT/trace_single.log" 2>&1
# the same one-stream command under the instruction counters: every detect_fused_kernel / describe_all_kernel launch of this
# pass is one the single-stream leg of bench.py times (same launch sequence, same chunk heights) -> profiles/valu.json
echo "progress: pmc pass (one stream)"; rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_single" -- python3 bench.py --steps 10 --warmup 3 --streams 1 --pyramid-in-detect 2 --legs single --profile-run $* > "$OUT/pmc_single.log" 2>&1
echo "progress: pmc pass"; rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $BENCH_ARGS > "$OUT/pmc_sq.log" 2>&1
echo "progress: pmc pass"; rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -- python3 bench.py $BENCH_ARGS > "$OUT/pmc_sq2.log" 2>&1
echo "progress: pmc pass"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $BENCH_ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "progress: pmc pass"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $BENCH_ARGS > "$OUT/pmc_write.log" 2>&1
echo "progress: pmc pass"; rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OUT/pmc_lds" -- python3 bench.py $BENCH_ARGS > "$OUT/pmc_lds.log" 2>&1
echo "progress: pmc pass"; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -- python3 bench.py $BENCH_ARGS > "$OUT/pmc_tcc.log" 2>&1
du -sh "$OUT"
