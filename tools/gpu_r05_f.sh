#!/bin/bash
# round 5, GPU call F: sampling -- the folded LayerNorm-GEMM (FD_LN_FOLD_WEIGHTS=1/0), heads writing the self-conditioning input,
# reverse step on the network's own score dtypes; trajectory parity tests
O=gpurun_out/r05f
mkdir -p $O
timeout 900 python -m pytest tests/test_ln_gemm.py tests/test_sampler.py tests/test_diffuser.py tests/test_parity_full.py -m gpu -x -q -k "ln_gemm or sampler or fold or traject or diffuser or advance or device_steps" > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -6 $O/gputest.log
grep "parity\]" $O/gputest.log | tail -8
for cfg in "128 1" "256 1"; do
  set -- $cfg
  for v in 1 0 1 0; do
    FD_LN_FOLD_WEIGHTS=$v timeout 300 python bench.py --mode sample --n-res $1 --batch $2 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$1 B=$2 fold=$v', d['value'], d['config']['ms_per_diffusion_step'])"
  done
done
for v in 1 0; do
  FD_LN_FOLD_WEIGHTS=$v timeout 300 python bench.py --mode sample --n-res 128 --batch 32 --steps 1 --warmup 1 --num-t 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=128 B=32 fold=$v', d['value'], d['config']['ms_per_diffusion_step'])"
done
