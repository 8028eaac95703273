#!/bin/bash
# round 6, GPU call T: 400 timed steps in one run, per-step times: how often a step stalls, and whether it is periodic
O=gpurun_out/r06t
mkdir -p $O
for i in 1 2; do
  FD_BENCH_STEP_TRACE=1 timeout 400 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-sampling 2> $O/long_err_$i.txt | tail -1 > $O/long_$i.json
  python - $O/long_err_$i.txt <<'PY'
import re, sys, statistics
t = [float(x) for x in re.search(r"in order\): (.*)", open(sys.argv[1]).read()).group(1).split()]
print(len(t), "steps: median", statistics.median(t), "mean", round(sum(t) / len(t), 3), "min", min(t), "steps above 22.5 ms:", [(i, x) for i, x in enumerate(t) if x > 22.5])
PY
done
