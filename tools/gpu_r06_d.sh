#!/bin/bash
# round 6, GPU call D: the key-side IPA backward kernel: parity, microbenchmark, the training step with / without it
O=gpurun_out/r06d
mkdir -p $O
timeout 900 python -m pytest tests/test_ipa_flash.py -m gpu -x -q -k "keys or bwd" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/bench_ipa_keys.py > $O/ipa_keys.log 2>&1; cat $O/ipa_keys.log
for i in 1 2; do
  for w in 1 0; do
    FD_IPA_FLASH_KEYS=$w timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_k${w}_$i.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06d/step_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 1200 python -m pytest tests/test_switches.py -m gpu -x -q > $O/tests2.log 2>&1; tail -3 $O/tests2.log
