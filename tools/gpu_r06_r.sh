#!/bin/bash
# round 6, GPU call R: fd_gemm tile 14 with two 16-k stages per ring slot (FD_GEMM_W_SUB=2: one barrier per 32 k) against one
O=gpurun_out/r06r
mkdir -p $O
FD_GEMM_W_SUB=2 timeout 600 python -m pytest tests/test_gemm_w.py tests/test_gemm.py -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for s in 1 2; do
  echo "== FD_GEMM_W_SUB=$s" >> $O/node_gemm.txt
  FD_GEMM_W_SUB=$s timeout 300 python tools/bench_node_gemm.py 2>/dev/null >> $O/node_gemm.txt
done
cat $O/node_gemm.txt | cut -c1-230
for i in 1 2 3; do
  for s in 1 2; do
    FD_GEMM_W_SUB=$s timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_sub${s}_$i.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06r/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
