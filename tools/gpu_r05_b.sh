#!/bin/bash
# round 5, GPU call B: the forward variants after the input-row reload (register pressure), same-box against the round-4 tree
O=gpurun_out/r05b
mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_edge_mlp.py tests/test_parity_full.py -m gpu -x -q -k "edge_mlp or benchmarked or n200 or n128_500" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -4 $O/gputest.log
timeout 300 python tools/bench_edge_mlp.py --shapes 30x128,8x512,1x128,1x256 > $O/edge_new.log 2>&1
(cd tools/probes/_r04_tree && timeout 300 python tools/bench_edge_mlp.py --shapes 30x128,8x512,1x128,1x256 > $R/$O/edge_old.log 2>&1)
cat $O/edge_new.log $O/edge_old.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_new_$i.json
  (cd tools/probes/_r04_tree && timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $R/$O/step_old_$i.json)
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/step_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline']['serialised_ms_per_step_by_kernel_class'])
    except Exception as e: print(f, 'ERR', e)
PY
for cfg in "128 1" "256 1" "512 8"; do
  set -- $cfg
  S=2; [ "$1" = "512" ] && S=1
  NT=500; [ "$1" = "512" ] && NT=60
  timeout 400 python bench.py --mode sample --n-res $1 --batch $2 --steps $S --warmup 1 --num-t $NT 2>/dev/null | tail -1 > $O/sample_new_n$1_b$2.json
  (cd tools/probes/_r04_tree && timeout 400 python bench.py --mode sample --n-res $1 --batch $2 --steps $S --warmup 1 --num-t $NT 2>/dev/null | tail -1 > $R/$O/sample_old_n$1_b$2.json)
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/sample_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['config']['ms_per_diffusion_step'])
    except Exception as e: print(f, 'ERR', e)
PY
