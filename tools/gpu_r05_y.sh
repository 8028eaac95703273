#!/bin/bash
# round 5, GPU call Y: the per-row IPA attention kernel with 512 threads (one wave per head) for launches with few query rows
# (FD_IPA_ATTN_THREADS=256 / 512 forces; default: 512 up to 512 rows) -- parity, lone-backbone sampling
timeout 300 python -m pytest tests/test_ipa_attn.py tests/test_ipa_flash.py -x -q -m gpu 2>&1 | tail -2
for n in 128 256; do
  for v in 256 512 256 512; do
    FD_IPA_ATTN_THREADS=$v timeout 300 python bench.py --mode sample --n-res $n --batch 1 --steps 1 --warmup 1 --num-t 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n B=1 threads=$v', d['config'].get('ms_per_diffusion_step'))"
  done
done
