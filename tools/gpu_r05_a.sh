#!/bin/bash
# round 5, GPU call A: the GPU test tier, the fused edge kernels and the training step against the round-4 tree on the same box
# (tools/probes/_r04_tree: `git archive` of the round-4 head + its library), sampling A/B, the default bench line
O=gpurun_out/r05a
mkdir -p $O
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout 300 python tools/bench_edge_mlp.py --shapes 30x128,8x512,1x128 > $O/edge_new.log 2>&1
(cd tools/probes/_r04_tree && timeout 300 python tools/bench_edge_mlp.py --shapes 30x128,8x512,1x128 > $R/$O/edge_old.log 2>&1)
cat $O/edge_new.log $O/edge_old.log
for i in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_new_$i.json
  (cd tools/probes/_r04_tree && timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $R/$O/step_old_$i.json)
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05a/step_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
    except Exception as e: print(f, 'ERR', e)
PY
for cfg in "128 1" "256 1"; do
  set -- $cfg
  timeout 300 python bench.py --mode sample --n-res $1 --batch $2 --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/sample_new_n$1_b$2.json
  (cd tools/probes/_r04_tree && timeout 300 python bench.py --mode sample --n-res $1 --batch $2 --steps 2 --warmup 1 2>/dev/null | tail -1 > $R/$O/sample_old_n$1_b$2.json)
  FD_SAMPLER_DEVICE_STEPS=0 FD_MERGE_SKIP=0 timeout 300 python bench.py --mode sample --n-res $1 --batch $2 --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/sample_newoff_n$1_b$2.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05a/sample_*.json')):
    try:
        d=json.loads(open(f).read()); print(f, d['value'], d['config']['ms_per_diffusion_step'])
    except Exception as e: print(f, 'ERR', e)
PY
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 1500 $O/bench_default.json
