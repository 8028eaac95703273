#!/bin/bash
# round 6, GPU call I: the one-launch sequence-attention backward: parity, microbenchmark, the training step with / without it
O=gpurun_out/r06i
mkdir -p $O
timeout 600 python -m pytest tests/test_seq_attn.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python tools/bench_seq_attn_bwd.py > $O/seq_attn_bwd.log 2>&1; cat $O/seq_attn_bwd.log
for i in 1 2; do
  for w in 1 0; do
    FD_SEQ_ATTN_BWD_FUSED=$w timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_s${w}_$i.json
  done
done
FD_SEQ_ATTN_BWD_FUSED=1 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_s1.json
FD_SEQ_ATTN_BWD_FUSED=0 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_s0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06i/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 1200 python -m pytest tests/test_switches.py -m gpu -x -q > $O/tests2.log 2>&1; tail -3 $O/tests2.log
