"""Timing-only ablations of the pre-split-weight GEMM (csrc/fd_gemm_w.h, GW_ABL bit mask; variant libraries built by
tools/probes/lib_variant.py fd_gemm gw<mask> -DFD_PROBE_BUILD -DGW_ABL=<mask>): 1 no loop loads, 2 no LDS writes, 4 no split VALU,
8 no MFMAs, 16 no epilogue, 32 no fragment reads.   python tools/probes/gemm_w_ablate.py   (GPU box)"""
import glob, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import hip
from tools.bench_node_gemm import timeit
dev = "cuda"
M = 3840
libs = {"shipped": hip.get_lib()}
for p in sorted(glob.glob(os.path.join(ROOT, "tools", "probes", "libfd_var_gw*.so")), key=lambda q: int(q.split("gw")[-1][:-3])):
    libs["abl " + p.split("gw")[-1][:-3]] = hip.FdLib(p)
shapes = [(1024, 1024, 12), (2176, 1024, 12), (6816, 256, 12), (320, 320, 14), (320, 320, 13), (1024, 1024, 13)]
print("us per launch; columns = " + " | ".join(libs))
for (N, K, tile) in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
    pl = torch.empty((3, W.numel()), dtype=torch.int16, device=dev)
    libs["shipped"].call("fd_split_planes", W, W.numel(), pl)
    row = []
    for name, lib in libs.items():
        row.append(timeit(lambda: lib.gemm(A, W, C, M, N, K, (K, 1), (1, K), N, tile=tile, b_planes=(pl.data_ptr(), W.numel()))))
    print(f"tile {tile} N={N:5d} K={K:5d}: " + " | ".join(f"{t:7.1f}" for t in row), flush=True)
