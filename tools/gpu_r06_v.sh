#!/bin/bash
# round 6, GPU call V: the [40, 128] gradient of IPA's [linear_b ; down_z] as one fd_pair_dw launch (FD_ZB_DW_STREAM) against the
# split-K fd_gemm + column-sum launches: parity, kernel times, training step
O=gpurun_out/r06v
mkdir -p $O
timeout 900 python -m pytest tests/test_pair_dw.py tests/test_abi.py -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 1200 python -m pytest tests/test_switches.py -m gpu -x -q -k "zb_dw or grouped_pair" > $O/tests2.log 2>&1; tail -2 $O/tests2.log
for i in 1 2 3; do
  for s in 1 0; do
    FD_ZB_DW_STREAM=$s timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_zb${s}_$i.json
  done
done
FD_ZB_DW_STREAM=1 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_zb1.json
FD_ZB_DW_STREAM=0 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/mixed_zb0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06v/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FD_BENCH_PROFILE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python bench.py --steps 5 --warmup 2 --no-sampling --no-cpu-baseline > $O/kt.log 2>&1
python tools/kernel_stats_md.py $O/kt/p_kernel_stats.csv "training step with FD_ZB_DW_STREAM=1" 2>/dev/null | grep -i "pair_dw\|colsum4\|true, false>\|gemm_kernel<64, 64, 2, 2, false, false, true" | cut -c1-160
find $O/kt -name "*.csv" -size +1M -delete
