#!/bin/bash
# round 5, GPU call D: the fused pair reductions of the edge backward -- tests, microbench, step A/B (FD_EDGE_PAIR_REDUCE=1/0)
O=gpurun_out/r05d
mkdir -p $O
timeout 900 python -m pytest tests/test_edge_mlp.py tests/test_switches.py tests/test_parity_full.py -m gpu -x -q -k "edge_mlp or switches_gpu or benchmarked or n200 or n256_b7" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -4 $O/gputest.log
timeout 300 python tools/bench_edge_mlp.py --shapes 30x128,12x200,7x256,2x512 2>&1 | grep -v amdgpu.ids | tee $O/edge.log
for i in 1 2; do
  for v in 1 0; do
    FD_EDGE_PAIR_REDUCE=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_pr${v}_$i.json
  done
done
FD_EDGE_PAIR_REDUCE=1 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 2>/dev/null | tail -1 > $O/mixed_pr1.json
FD_EDGE_PAIR_REDUCE=0 timeout 300 python bench.py --mixed-n --steps 12 --warmup 3 2>/dev/null | tail -1 > $O/mixed_pr0.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05d/*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
