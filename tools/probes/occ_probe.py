"""co-residency probe: the same tile with <= 1 block per CU and with 2 / 4 blocks per CU"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from se3_diffusion_amd import hip
from tools.bench_node_gemm import timeit
lib = hip.get_lib(); dev = "cuda"
M = 3840
K = 1024
for tile, ns in ((12, (1024, 2048, 4096)), (13, (512, 1024, 2048, 4096)), (14, (256, 512, 1024, 2048))):
    for N in ns:
        A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
        pl = torch.empty((3, W.numel()), dtype=torch.int16, device=dev); lib.call("fd_split_planes", W, W.numel(), pl)
        t = timeit(lambda: lib.gemm(A, W, C, M, N, K, (K, 1), (1, K), N, tile=tile, b_planes=(pl.data_ptr(), W.numel())))
        bm = 128 if tile == 12 else 64
        bn = 64 if tile == 14 else 128
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        print(f"tile {tile} K={K} N={N}: {t:7.1f} us  blocks={tiles} per-CU={tiles/256:.2f}  TF={2.0*M*N*K/t/1e6:.1f}")
