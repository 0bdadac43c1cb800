"""The legs of bench.py (the driver's benchmark command), one module each.  bench.py itself parses the arguments, starts
the ranks, runs the timed region (bench_legs.timed) and calls the legs in a fixed order; every leg takes the shared `Run`
object (bench_legs.common) and adds its keys to `run.out`, the ONE JSON line rank 0 prints.

  launch   spawn_ranks / dry_launch: `python bench.py --gpus N` as a plain command, and its CPU rehearsal over gloo
  models   gather_model / tiled_model: committed PREDICTIONS of the multi-GPU steps (no multi-GPU node was reachable)
  timed    inputs, the exchange, pre-flight, the W + K steps of `value`
  repeat   the timed region again (spread, from idle, the pyramid policy A/B)
  single   one stream with HIP events per launch: stage table, VALU rooflines of the two kernels that own the step
  two_stage  the reference's LaplaceMulti -> DoG in HBM -> FindPointsMulti pipeline: `roofline` (the north-star gate)
  host     SiftData made host-visible; host-to-host (upload every step); the C ABI's cusift_pipe_*
  content  other image content, initBlur = 0, ragged widths
  configs  BASELINE configs[0], [1] and [4] (`config_legs`): the 640x480 fixture, one 1080p frame, one 8192^2 image --
           whole on one GPU and, with N > 1, strip-tiled over the ranks that are up
  match    MatchSiftData on the fp32 MFMA
  cpu      `cpu_baseline`: the CPU oracle (tests/oracle_binding.py) on the host cores; OpenCV if importable
"""
