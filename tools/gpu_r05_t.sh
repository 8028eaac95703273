#!/bin/bash
timeout 300 python -m pytest tests/test_gemm_direct.py -x -q -m gpu 2>&1 | tail -2
cat > /tmp/shapes.py <<'PY'
import torch, sys, os
sys.path.insert(0, '.')
from se3_diffusion_amd import hip
lib = hip.get_lib()
def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M in (64, 128, 256):
    out = []
    for (N, K) in ((256, 2688), (320, 1280), (320, 960), (320, 640), (320, 320)):
        A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda'); C = torch.empty(M, N, device='cuda')
        out.append(f"{N}x{K}: {timeit(lambda: lib.gemm(A, W, C, M, N, K, (K, 1), (1, K), N, bias=b)):5.1f}")
    print("waves=" + os.environ.get("FD_GEMM_DIRECT_T16_WAVES", "4"), f"M={M:4d} ", "  ".join(out))
PY
for w in 4 8; do FD_GEMM_DIRECT_T16_WAVES=$w python /tmp/shapes.py 2>&1 | grep -v amdgpu; done
for n in 128 256; do
  for v in 4 8 4 8; do
    FD_GEMM_DIRECT_T16_WAVES=$v timeout 300 python bench.py --mode sample --n-res $n --batch 1 --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n B=1 T16_WAVES=$v', d['value'], d['config'].get('ms_per_diffusion_step'))"
  done
done
