"""Microbenchmark: the grouped pair-row weight-gradient kernel (fd_pair_dw) against the fd_gemm launches it replaces
(dW2 384x384, dW1z 384x128, dWf 128x384, dWfz 128x128 at B*N*N rows).  python tools/bench_pair_dw.py [rows] [blocks]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import ops  # noqa: E402

mv = lambda t: (t, 0, t.stride(0))


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 30 * 128 * 128
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    d2, h1, d1, z, h2, dy = rn(rows, 384), rn(rows, 384), rn(rows, 384), rn(rows, 128), rn(rows, 384), rn(rows, 128)
    gW2, gb2, gW1, gWf = (torch.zeros(384, 384, device=dev), torch.zeros(384, device=dev), torch.zeros(384, 384, device=dev),
                          torch.zeros(128, 384, device=dev))

    def grouped(nitems=5):
        items = [dict(A=(d2, 0, 384), B=(h1, 128 * j, 384), C=(gW2, 128 * j, 384), colsum=gb2 if j == 0 else None) for j in range(3)]
        items.append(dict(A=(d1, 0, 384), B=(z, 0, 128), C=(gW1, 0, 384)))
        items.append(dict(A=(h2, 0, 384), A_add=(z, 0, 128), B=(dy, 0, 128), C=(gWf, 0, 384), trans=True))
        ops.pair_dw(items[:nitems], rows, blocks=blocks)

    def unfused():
        ops.linear_dw(mv(d2), mv(h1), mv(gW2), rows, 384, 384, db=gb2)
        ops.linear_dw(mv(d1), mv(z), (gW1, 0, 384), rows, 384, 128)
        ops.linear_dw(mv(dy), mv(h2), mv(gWf), rows, 128, 384)
        ops.linear_dw(mv(dy), mv(z), (gWf, 0, 384), rows, 128, 128)

    def timeit(fn, n=10):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    flops5 = 2.0 * rows * 384 * 128 * 5
    flops_all = 2.0 * rows * (384 * 384 + 2 * 384 * 128 + 128 * 128)
    for name, fn, fl in (("fd_pair_dw, 5 items", grouped, flops5), ("fd_pair_dw, 3 items (dW2)", lambda: grouped(3), flops5 * 0.6),
                         ("fd_gemm x 4 (dW2, dW1z, dWf, dWfz)", unfused, flops_all)):
        ms = timeit(fn)
        print(f"{name:40s} rows={rows}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
    # the edge embedder's three 128-wide layers: 128 x 128 items / fd_gemm (64 x 64 fp32 tiles)
    dh = [rn(rows, 128) for _ in range(3)]
    xs = [rn(rows, 128), rn(rows, 128), rn(rows, 120)]
    eW = [torch.zeros(128, x.shape[1], device=dev) for x in xs]
    eb = [torch.zeros(128, device=dev) for _ in xs]
    bands = [dict(A=(dh[i], 0, 128), B=(xs[i], 0, xs[i].shape[1]), C=(eW[i], 0, xs[i].shape[1]), colsum=eb[i],
                  b_cols=0 if xs[i].shape[1] == 128 else xs[i].shape[1]) for i in range(3)]
    narrow = [dict(b, a_bands=1) for b in bands]
    fl_e = 2.0 * rows * 128 * (128 + 128 + 120)
    for name, fn in (("fd_pair_dw, 3 items of 128 x 128", lambda: ops.pair_dw(narrow, rows, blocks=blocks)),
                     ("fd_gemm x 3 (embedder)", lambda: [ops.linear_dw(mv(dh[i]), mv(xs[i]), mv(eW[i]), rows, 128, xs[i].shape[1], db=eb[i]) for i in range(3)])):
        ms = timeit(fn)
        print(f"{name:40s} rows={rows}  {ms:7.3f} ms  {fl_e / ms / 1e9:7.1f} TFLOP/s", flush=True)
    # agreement of the two paths (both accumulate 12 runs)
    for t in (gW2, gb2, gW1, gWf):
        t.zero_()
    grouped()
    a = [t.clone() for t in (gW2, gb2, gW1[:, :128], gWf)]
    for t in (gW2, gb2, gW1, gWf):
        t.zero_()
    unfused()
    torch.cuda.synchronize()
    for x, y, n in zip(a, (gW2, gb2, gW1[:, :128], gWf), ("W2", "b2", "W1z", "Wf")):
        print(f"  {n}: max |grouped - gemm| / max = {float((x - y).abs().max() / y.abs().max()):.2e}")


if __name__ == "__main__":
    main()
