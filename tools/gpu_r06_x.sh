#!/bin/bash
# round 6, GPU call X: the edge embedder's backward in row chunks (its weight gradients leave the exposed tail of the step), with / without
# the streaming [linear_b ; down_z] gradient kernel
O=gpurun_out/r06x
mkdir -p $O
timeout 900 python -m pytest tests/test_edge_embed_bwd.py tests/test_ipa_zb_dw.py -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2 3; do
  for cfg in "1 0" "2 0" "2 1" "3 0" "4 0"; do
    set -- $cfg
    FD_EMBED_BWD_CHUNKS=$1 FD_ZB_DW_STREAM=$2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_c$1_zb$2_$i.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06x/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['config']['step_ms_spread']['median'])
    except Exception as e: print(f, 'ERR', e)
PY
