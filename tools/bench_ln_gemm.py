"""fd_ln_gemm (LayerNorm inside the consuming GEMM, sampling) against the plain latency GEMM of the same shape, us per launch.
   python tools/bench_ln_gemm.py [M ...]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import ops  # noqa: E402
from se3_diffusion_amd.ops import mv  # noqa: E402


def timeit(fn, reps=50, warm=5):
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
    dev = "cuda"
    for M in ([int(a) for a in sys.argv[1:]] or [128, 256]):
        for (N, K, ln_out) in ((960, 320, False), (320, 320, False), (320, 320, True), (1024, 256, True), (256, 320, True)):
            x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
            g, bt = torch.randn(K, device=dev), torch.randn(K, device=dev)
            out, lo = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
            t0 = timeit(lambda: ops.linear(mv(x), mv(W), b, mv(out), M, N, K))
            t1 = timeit(lambda: ops.ln_linear(mv(x), g, bt, mv(W), b, mv(out), M, N, K, ln_out=mv(lo) if ln_out else None))
            t2 = timeit(lambda: ops.layernorm(mv(x), g, bt, mv(lo), M, K))
            print(f"M={M:4d} N={N:5d} K={K:4d} ln_out={int(ln_out)}: plain GEMM {t0:5.1f} us   LN-GEMM {t1:5.1f} us   LayerNorm alone {t2:5.1f} us")


if __name__ == "__main__":
    main()
