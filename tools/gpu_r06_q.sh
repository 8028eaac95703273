#!/bin/bash
# round 6, GPU call Q: the 8-wave fd_group_dw beside the main stream on fewer CUs (FD_NODE_DW_BLOCKS), against the four-wave form
O=gpurun_out/r06q
mkdir -p $O
for i in 1 2; do
  for b in 256 192 160 128; do
    FD_GROUP_DW_LOCKSTEP=0 FD_NODE_DW_BLOCKS=$b timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_nolock_b${b}_$i.json
  done
  FD_NODE_DW_BLOCKS=192 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_lock_b192_$i.json
  FD_GROUP_DW_V1=1 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_v1_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06q/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
