#!/bin/bash
# round 6, GPU call P: fd_group_dw, 8-wave forms with measured stage costs (3 : 2) against the four-wave form: kernel time per item list, parity, step
O=gpurun_out/r06p
mkdir -p $O
for v in lock nolock v1; do
  unset FD_GROUP_DW_V1 FD_GROUP_DW_LOCKSTEP
  if [ $v = v1 ]; then export FD_GROUP_DW_V1=1; fi
  if [ $v = nolock ]; then export FD_GROUP_DW_LOCKSTEP=0; fi
  echo "== $v" >> $O/items.txt
  timeout 300 python tools/bench_group_dw_items.py 2>/dev/null >> $O/items.txt
done
unset FD_GROUP_DW_V1 FD_GROUP_DW_LOCKSTEP
cat $O/items.txt
FD_GROUP_DW_LOCKSTEP=0 timeout 600 python -m pytest tests/test_group_dw.py -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2 3; do
  FD_GROUP_DW_LOCKSTEP=0 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_nolock_$i.json
  FD_GROUP_DW_V1=1 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_v1_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06p/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
