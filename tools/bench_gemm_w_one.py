"""One node-level GEMM shape on one tile, a few launches (for rocprofv3 --pmc):  python tools/bench_gemm_w_one.py M N K tile fwd|dx [ksplit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import hip  # noqa: E402

M, N, K, tile = (int(a) for a in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "fwd"
ks = int(sys.argv[6]) if len(sys.argv) > 6 else 1
lib = hip.get_lib()
dev = "cuda"
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) if mode == "fwd" else torch.randn(K, N, device=dev)
C = torch.zeros(M, N, device=dev)
pl = torch.empty((3, W.numel()), dtype=torch.int16, device=dev)
lib.call("fd_split_planes", W, W.numel(), pl)
b_str = (1, K) if mode == "fwd" else (N, 1)
for _ in range(6):
    lib.gemm(A, W, C, M, N, K, (K, 1), b_str, N, tile=tile, ksplit=ks, b_planes=(pl.data_ptr(), W.numel()))
torch.cuda.synchronize()
