"""ring depth of the pre-split-weight GEMM (csrc/fd_gemm_w.h, -DGW_RING=n variants built by tools/probes/lib_variant.py)"""
import glob, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import hip
from tools.bench_node_gemm import timeit
dev = "cuda"
M = 3840
libs = {"ring 3": hip.get_lib()}
for p in sorted(glob.glob(os.path.join(ROOT, "tools", "probes", "libfd_var_ring*.so"))):
    libs["ring " + p.split("ring")[-1][:-3]] = hip.FdLib(p)
shapes = [(1024, 1024, 12, 1), (2176, 1024, 12, 1), (6816, 256, 12, 1), (320, 320, 14, 1), (256, 256, 14, 1), (320, 320, 13, 1), (960, 320, 12, 1),
          (256, 2688, 14, 1), (256, 2688, 12, 4), (2688, 256, 13, 1), (1024, 1024, 13, 1), (1024, 1024, 14, 1)]
print("us per launch (fwd layout); columns = " + " | ".join(libs))
for (N, K, tile, ks) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.zeros(M, N, device=dev)
    pl = torch.empty((3, W.numel()), dtype=torch.int16, device=dev)
    libs["ring 3"].call("fd_split_planes", W, W.numel(), pl)
    row = []
    for name, lib in libs.items():
        row.append(timeit(lambda: lib.gemm(A, W, C, M, N, K, (K, 1), (1, K), N, tile=tile, ksplit=ks, b_planes=(pl.data_ptr(), W.numel()))))
    print(f"tile {tile} N={N:5d} K={K:5d} ks={ks}: " + " | ".join(f"{t:7.1f}" for t in row), flush=True)
