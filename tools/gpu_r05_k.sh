#!/bin/bash
# round 5, GPU call K: non-temporal LOADS of the streams that are read once -- pair_dw's operands (dw_nt), the fused edge kernels'
# upstream gradient / LayerNorm output / residual re-read (nt_in), both -- against the shipped library, alternating; the training step
cp se3_diffusion_amd/lib/libfd_hip.so /tmp/base.so
run() { timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; }
for i in 1 2; do
  for v in shipped dw_nt nt_in nt_both; do
    if [ $v = shipped ]; then cp /tmp/base.so se3_diffusion_amd/lib/libfd_hip.so; else cp tools/probes/libfd_ev_$v.so se3_diffusion_amd/lib/libfd_hip.so; fi
    run $v
  done
done
cp /tmp/base.so se3_diffusion_amd/lib/libfd_hip.so
timeout 200 python tools/bench_pair_dw.py 2>&1 | grep -v amdgpu | tail -8
cp tools/probes/libfd_ev_dw_nt.so se3_diffusion_amd/lib/libfd_hip.so
timeout 200 python tools/bench_pair_dw.py 2>&1 | grep -v amdgpu | tail -8
cp /tmp/base.so se3_diffusion_amd/lib/libfd_hip.so
