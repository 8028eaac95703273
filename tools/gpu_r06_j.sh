#!/bin/bash
# round 6, GPU call J: the training step with the weight-gradient launches of the side stream left out (upper bound of what a
# cheaper fd_group_dw / fd_pair_dw can give the step)
O=gpurun_out/r06j
mkdir -p $O
for i in 1 2; do
  for s in none group pair both; do
    SKIP=$s timeout 300 python tools/probes/skip_dw.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_${s}_$i.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06j/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
