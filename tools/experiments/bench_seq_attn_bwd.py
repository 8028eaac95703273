"""fd_seq_attn_bwd (two launches) against the five launches it replaces, us per transformer layer.  python tools/bench_seq_attn_bwd.py [B N]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd.ops import lib  # noqa: E402

TH, THD, TD = 4, 80, 320


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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = "cuda"
    L = lib()
    R = B * N
    qkv = torch.randn(R, 3 * TD, device=dev); do = torch.randn(R, TD, device=dev)
    out = torch.empty(R, TD, device=dev); A = torch.empty(B, TH, N, N, device=dev)
    sc = 1.0 / math.sqrt(THD)
    L.call("fd_seq_attn_fwd", qkv, None, out, A, sc, B, N)
    dA = torch.empty(B, TH, N, N, device=dev); dqkv = torch.empty(R, 3 * TD, device=dev)

    def seq():
        L.gemm(do, qkv, dA, N, N, THD, (TD, 1), (1, 3 * TD), N, b_off=2 * TD, batch=B * TH, bdiv=TH,
               a_bs=(N * TD, THD), b_bs=(N * 3 * TD, THD), c_bs=(TH * N * N, N * N))
        L.gemm(A, do, dqkv, N, THD, N, (1, N), (TD, 1), 3 * TD, c_off=2 * TD, batch=B * TH, bdiv=TH,
               a_bs=(TH * N * N, N * N), b_bs=(N * TD, THD), c_bs=(N * 3 * TD, THD))
        L.call("fd_row_softmax_bwd", A, dA, B * TH * N, N)
        L.gemm(dA, qkv, dqkv, N, THD, N, (N, 1), (3 * TD, 1), 3 * TD, b_off=TD, batch=B * TH, bdiv=TH,
               a_bs=(TH * N * N, N * N), b_bs=(N * 3 * TD, THD), c_bs=(N * 3 * TD, THD), alpha=sc)
        L.gemm(dA, qkv, dqkv, N, THD, N, (1, N), (3 * TD, 1), 3 * TD, c_off=TD, batch=B * TH, bdiv=TH,
               a_bs=(TH * N * N, N * N), b_bs=(N * 3 * TD, THD), c_bs=(N * 3 * TD, THD), alpha=sc)
    t0 = timeit(seq)
    ref = dqkv.clone()
    t1 = timeit(lambda: L.call("fd_seq_attn_bwd", qkv, A, do, out, dA, dqkv, sc, B, N))
    err = float((dqkv - ref).abs().max() / ref.abs().max())
    print(f"B={B} N={N}: five launches {t0:.1f} us | fd_seq_attn_bwd {t1:.1f} us | maxdiff {err:.1e}")


if __name__ == "__main__":
    main()
