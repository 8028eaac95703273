#!/bin/bash
# round 6, GPU call G: the shared-operand key-side IPA backward in the training step (same box, alternating)
O=gpurun_out/r06g
mkdir -p $O
for i in 1 2; do
  for w in 1 0; do
    FD_IPA_FLASH_KEYS=$w timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_k${w}_$i.json
  done
done
FD_IPA_FLASH_KEYS=1 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_k1.json
FD_IPA_FLASH_KEYS=0 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_k0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06g/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
