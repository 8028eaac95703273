#!/bin/bash
# round 6, GPU call W: fd_ipa_zb_dw (the [40, 128] gradient of IPA's [linear_b ; down_z] as one light streaming launch): parity, the launch
# against what it replaces, training step and mixed lengths with / without it
O=gpurun_out/r06w
mkdir -p $O
timeout 900 python -m pytest tests/test_ipa_zb_dw.py tests/test_abi.py tests/test_empty.py -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 300 python tools/bench_zb_dw.py 2>/dev/null | tee $O/zb_dw_microbench.log
for i in 1 2 3; do
  for s in 1 0; do
    FD_ZB_DW_STREAM=$s timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_zb${s}_$i.json
  done
done
FD_ZB_DW_STREAM=1 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_zb1.json
FD_ZB_DW_STREAM=0 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_zb0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06w/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 1500 python -m pytest tests/test_switches.py tests/test_parity_full.py -m gpu -x -q -k "switch or benchmarked or mixed" > $O/tests2.log 2>&1; tail -2 $O/tests2.log
