#!/bin/bash
# round 6, GPU call C: node-level GEMMs on pre-split weight planes inside the training step: same-box A/B (FD_WEIGHT_PLANES=1/0),
# the GEMM launches by shape, the step-level parity tests
O=gpurun_out/r06c
mkdir -p $O
for i in 1 2; do
  for w in 1 0; do
    FD_WEIGHT_PLANES=$w timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_w${w}_$i.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06c/step_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 300 python tools/gemm_shapes.py 30 128 > $O/gemm_shapes.log 2>&1; head -40 $O/gemm_shapes.log
timeout 1200 python -m pytest tests/test_switches.py tests/test_gemm_w.py tests/test_module.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 1200 python -m pytest tests/test_parity_full.py -m gpu -x -q -k "benchmarked or n200 or n256_b7" > $O/tests2.log 2>&1; tail -3 $O/tests2.log
