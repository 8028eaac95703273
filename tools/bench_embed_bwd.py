"""The edge embedder's backward at the training shape (B=30 x N=128 -> 491,520 pair rows): fd_edge_embed_bwd + fd_group_dw
against the launch sequence they replace (fd_layernorm_bwd, two gated dX GEMMs, three split-K weight-gradient GEMMs)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import ops  # noqa: E402
from se3_diffusion_amd.ops import mv  # noqa: E402


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 30 * 128 * 128
    dev, C = "cuda", 128
    ops.set_grad_stream(False)
    rn = lambda *s: torch.randn(*s, device=dev)
    W2, W4, gamma = rn(C, C) / 8, rn(C, C) / 8, 1 + 0.1 * rn(C)
    h1, h2, h3, dy, x = torch.relu(rn(rows, C)), torch.relu(rn(rows, C)), rn(rows, C), rn(rows, C), rn(rows, 120)
    mean, rstd, rs = h3.mean(-1), 1 / torch.sqrt(h3.var(-1, unbiased=False) + 1e-5), torch.ones(rows, device=dev)
    d3, d2, d1 = (torch.empty(rows, C, device=dev) for _ in range(3))
    gW4, gW2, gW0 = torch.zeros(C, C, device=dev), torch.zeros(C, C, device=dev), torch.zeros(C, 120, device=dev)
    gb4, gb2, gb0, dg, db = (torch.zeros(C, device=dev) for _ in range(5))
    img = ops.edge_embed_bwd_pack(W2, W4)
    items = [(mv(d3), mv(h2), mv(gW4), gb4, C, C), (mv(d2), mv(h1), mv(gW2), gb2, C, C), (mv(d1), mv(x), mv(gW0), gb0, C, 120)]

    def fused_dx():
        ops.edge_embed_bwd(dy, h3, mean, rstd, gamma, rs, h2, h1, img, d3, d2, d1, dg, db, rows)

    def unfused_dx():
        ops.layernorm_bwd(mv(dy), mv(h3), gamma, mean, rstd, mv(d3), rows, C, rowscale=rs, dgamma=dg, dbeta=db)
        ops.linear_dx(mv(d3), mv(W4), mv(d2), rows, C, C, gate=mv(h2))
        ops.linear_dx(mv(d2), mv(W2), mv(d1), rows, C, C, gate=mv(h1))

    def unfused_dw():
        ops.linear_dw(mv(d3), mv(h2), mv(gW4), rows, C, C, db=gb4)
        ops.linear_dw(mv(d2), mv(h1), mv(gW2), rows, C, C, db=gb2)
        ops.linear_dw(mv(d1), mv(x), mv(gW0), rows, C, 120, db=gb0)

    print(f"rows = {rows}")
    print(f"fd_edge_embed_bwd (LN bwd + 2 dX)            {timeit(fused_dx):7.3f} ms")
    print(f"fd_layernorm_bwd + 2 gated dX GEMMs          {timeit(unfused_dx):7.3f} ms")
    for b in (0, 256, 1024):
        print(f"fd_group_dw, 3 items, blocks={b or 512:4d}           {timeit(lambda: ops.group_dw(items, rows, blocks=b)):7.3f} ms")
    print(f"fd_gemm x 3 (split-K weight gradients)       {timeit(unfused_dw):7.3f} ms")


if __name__ == "__main__":
    main()
