#!/bin/bash
# round 5, GPU call G: shape threshold of the fused edge kernels on the new code (FD_EDGE_SHAPE=4/8 at the sizes around 65,536 rows)
for cfg in "256 1" "192 1" "128 4" "128 8"; do
  set -- $cfg
  for v in 4 8 4 8; do
    FD_EDGE_SHAPE=$v timeout 300 python bench.py --mode sample --n-res $1 --batch $2 --steps 1 --warmup 1 --num-t 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$1 B=$2 shape=$v', d['value'], d['config']['ms_per_diffusion_step'])"
  done
done
