#!/bin/bash
# round 6, GPU call M: fd_group_dw on fd_pair_dw's 8-wave block (384 x 128 / 128 x 128 units) against the four-wave 128 x 128 form
# (FD_GROUP_DW_V1=1): parity, the launch alone, the training step
O=gpurun_out/r06m
mkdir -p $O
timeout 600 python -m pytest tests/test_group_dw.py tests/test_pair_dw.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for v in new nolock v1; do
  echo "== $v" >> $O/group_dw.txt
  unset FD_GROUP_DW_V1 FD_GROUP_DW_LOCKSTEP; if [ $v = v1 ]; then export FD_GROUP_DW_V1=1; fi; if [ $v = nolock ]; then export FD_GROUP_DW_LOCKSTEP=0; fi
  timeout 200 python tools/bench_group_dw.py 3840 0 2>/dev/null >> $O/group_dw.txt
  timeout 200 python tools/bench_group_dw.py 3840 192 2>/dev/null | head -1 >> $O/group_dw.txt
  timeout 200 python tools/bench_group_dw.py 3840 128 2>/dev/null | head -1 >> $O/group_dw.txt
done
unset FD_GROUP_DW_V1 FD_GROUP_DW_LOCKSTEP
cat $O/group_dw.txt
timeout 200 python tools/bench_pair_dw.py > $O/pair_dw.txt 2>&1; tail -5 $O/pair_dw.txt
for i in 1 2; do
  FD_GROUP_DW_LOCKSTEP=0 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_nolock_$i.json
  FD_GROUP_DW_V1=1 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_v1_$i.json
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_new_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06m/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
