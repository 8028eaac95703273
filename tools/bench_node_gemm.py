"""Node-level GEMM shapes of the training step (M = B*N = 3840 rows) on every tile that accepts them: where does the latency
kernel (tile 5) stop winning?   python tools/bench_node_gemm.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import hip  # noqa: E402


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    lib = hip.get_lib()
    dev = "cuda"
    rows = [int(a) for a in sys.argv[1:]] or [3840, 1920, 1024]
    shapes = [(320, 320), (256, 256), (960, 320), (320, 960), (1280, 320), (320, 1280), (256, 2688), (2688, 256), (6816, 256), (256, 6816)]
    for M in rows:
        print(f"M = {M}: us per launch (replayed back to back), activations [M,K] x weights [N,K]^T + bias")
        for (N, K) in shapes:
            A = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev)
            b = torch.randn(N, device=dev)
            C = torch.empty(M, N, device=dev)
            line = [f"  N={N:5d} K={K:5d}"]
            for tile in (0, 2, 10, 5, 4):
                try:
                    t = timeit(lambda: lib.gemm(A, W, C, M, N, K, (K, 1), (1, K), N, bias=b, tile=tile))
                    line.append(f"tile {tile}: {t:7.1f}")
                except Exception as e:  # noqa: BLE001
                    line.append(f"tile {tile}:     n/a")
            print(" | ".join(line), flush=True)


if __name__ == "__main__":
    main()
