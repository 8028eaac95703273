#!/bin/bash
# round 6, GPU call Y: the main stream waiting for the side stream in front of the pair-level kernels of the backward (no two pair-level
# kernels side by side) against the shipped free-running overlap
O=gpurun_out/r06y
mkdir -p $O
for i in 1 2; do
  for w in none edge embed both; do
    WAIT=$w timeout 300 python tools/probes/side_wait.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_${w}_$i.json
  done
  WAIT=edge FD_PAIR_DW_BLOCKS=256 timeout 300 python tools/probes/side_wait.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_edge_b256_$i.json
  WAIT=both FD_PAIR_DW_BLOCKS=256 timeout 300 python tools/probes/side_wait.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_both_b256_$i.json
  WAIT=both FD_PAIR_DW_BLOCKS=208 timeout 300 python tools/probes/side_wait.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_both_b208_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06y/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['config']['step_ms_spread']['median'])
    except Exception as e: print(f, 'ERR', e)
PY
