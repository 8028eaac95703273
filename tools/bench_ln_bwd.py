"""fd_layernorm_bwd at the node-level shapes of the training step: with / without the gamma / beta gradients (atomics).
    python tools/bench_ln_bwd.py      (GPU box)"""
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
    for R, C in ((3840, 320), (3840, 256), (1920, 320), (128, 320)):
        x = torch.randn(R, C, device=dev); dy = torch.randn(R, C, device=dev); g = torch.randn(C, device=dev)
        mean = x.mean(-1).contiguous(); rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()
        dx = torch.empty(R, C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        t0 = timeit(lambda: ops.layernorm_bwd(mv(dy), mv(x), g, mean, rstd, mv(dx), R, C, dgamma=dg, dbeta=db))
        t1 = timeit(lambda: ops.layernorm_bwd(mv(dy), mv(x), g, mean, rstd, mv(dx), R, C))
        print(f"rows={R} C={C}: with dgamma/dbeta {t0:.1f} us | dx only {t1:.1f} us", flush=True)


if __name__ == "__main__":
    main()
