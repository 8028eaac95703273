#!/bin/bash
# round 5, GPU call L: the column-split edge kernel for a lone backbone (csrc/fd_edge_mlp_pair.hip): parity, the launch alone
# (FD_EDGE_PAIR=0 = the 4-wave shape it replaces), N=128 / N=64 B=1 sampling with and without it
timeout 300 python -m pytest tests/test_edge_mlp.py -x -q -m gpu -k pair 2>&1 | tail -3
for pr in 0 1 0 1; do
  echo "FD_EDGE_PAIR=$pr"; FD_EDGE_PAIR=$pr timeout 200 python tools/bench_edge_mlp.py --shapes 1x128,1x96,1x64 --fwd-only 2>&1 | grep -v amdgpu | tail -4
done
for pr in 0 1 0 1; do
  FD_EDGE_PAIR=$pr timeout 300 python bench.py --mode sample --n-res 128 --batch 1 --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=128 B=1 FD_EDGE_PAIR=$pr', d['value'], d['config'].get('ms_per_diffusion_step'))"
done
