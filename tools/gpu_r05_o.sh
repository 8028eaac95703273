#!/bin/bash
# round 5, GPU call O: the time series of the training step inside one process (the priming chunks of the profile run went from 21.5 to
# 20.98 ms after ~1 s of load, the 10 timed steps behind them measured 21.3): per-step HIP-event times of longer timed regions
for k in 10 120; do
  FD_BENCH_STEP_TRACE=1 timeout 300 python bench.py --steps $k --warmup 3 --no-cpu-baseline --no-sampling 2> gpurun_out/o_$k.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps=$k', d['ms_per_step'], d['priming_chunk_ms'], d['config']['step_ms_spread'])"
  grep "per-step" gpurun_out/o_$k.err
done
