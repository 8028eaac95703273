#!/bin/bash
# round 5, GPU call C: two forms of the forward's input-row handling (product: two 16-blocks per k-step fetched one k-step ahead;
# "reload": the whole row fetched again per chunk) on the four launch kinds, four sizes; outputs bit-compared
O=gpurun_out/r05c
mkdir -p $O
export EDGE_VARIANTS_EXTRA="reload:prebuilt"
for cfg in "30 128" "8 512" "1 128" "1 256" "12 200"; do
  set -- $cfg
  timeout 300 python tools/probes/edge_variants.py --rows-b $1 --n $2 2>&1 | grep -v amdgpu.ids | tee -a $O/edge_variants.log
done
timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 > $O/step_new.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05c/step_new.json').read()); print('step', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
for cfg in "128 1" "512 8"; do
  set -- $cfg
  S=2; [ "$1" = "512" ] && S=1
  NT=500; [ "$1" = "512" ] && NT=60
  timeout 400 python bench.py --mode sample --n-res $1 --batch $2 --steps $S --warmup 1 --num-t $NT 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sample', d['config']['workload'][:60], d['value'], d['config']['ms_per_diffusion_step'])"
done
