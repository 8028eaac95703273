#!/bin/bash
# round 6, GPU call L: fd_group_dw with one row slice per XCD (FD_GROUP_DW_SLICES=8) against the tile-major order: parity, launch, step
O=gpurun_out/r06l
mkdir -p $O
FD_GROUP_DW_SLICES=8 timeout 600 python -m pytest tests/test_group_dw.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for s in 1 8 4 2; do
  echo "FD_GROUP_DW_SLICES=$s" >> $O/group_dw.txt
  FD_GROUP_DW_SLICES=$s timeout 200 python tools/bench_group_dw.py 3840 0 2>/dev/null | grep -v "^fd_gemm" >> $O/group_dw.txt
  FD_GROUP_DW_SLICES=$s FD_GROUP_DW_DEBUG=7 timeout 200 python tools/bench_group_dw.py 3840 0 2>/dev/null | head -1 >> $O/group_dw.txt
done
cat $O/group_dw.txt
for i in 1 2; do
  for s in 1 8; do
    FD_GROUP_DW_SLICES=$s timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_s${s}_$i.json
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06l/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
