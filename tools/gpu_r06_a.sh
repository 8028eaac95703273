#!/bin/bash
# round 6, GPU call A: baselines on this round's first box before any kernel work -- the node-level GEMM table, the GEMM launches of
# a training step by shape, the default training line
O=gpurun_out/r06a
mkdir -p $O
timeout 300 python tools/bench_node_gemm.py 3840 > $O/node_gemm.log 2>&1
cat $O/node_gemm.log
timeout 300 python tools/gemm_shapes.py 30 128 > $O/gemm_shapes.log 2>&1
cat $O/gemm_shapes.log
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/step.json').read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])
PY
