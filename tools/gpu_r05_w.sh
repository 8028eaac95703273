#!/bin/bash
# round 5, GPU call W: the embedders' first layers on features padded to K = 72 (options.embed_first_padded) -- parity (network-level
# comparison, trajectories against the reference goldens), lone-backbone and batched sampling with and without it
timeout 600 python -m pytest tests/test_ln_gemm.py tests/test_sampler.py "tests/test_parity_full.py::test_reference_trajectory_n128" tests/test_network.py -x -q -m gpu 2>&1 | tail -3
for cfg in "128 1" "256 1" "128 8"; do
  set -- $cfg
  for v in 0 1 0 1; do
    FD_EMBED_FIRST_PADDED=$v timeout 300 python bench.py --mode sample --n-res $1 --batch $2 --steps 1 --warmup 1 --num-t 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$1 B=$2 FD_EMBED_FIRST_PADDED=$v', d['value'], d['config'].get('ms_per_diffusion_step'))"
  done
done
