"""Sequence-transformer attention backward: fd_seq_attn_bwd (one launch) against the five launches it replaces (two batched GEMMs,
fd_row_softmax_bwd, two batched GEMMs).   python tools/bench_seq_attn_bwd.py [BxN ...]   (GPU box)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import hip  # noqa: E402
from tools.bench_node_gemm import timeit  # noqa: E402

TH, THD, TD = 4, 80, 320


def main():
    L = hip.get_lib()
    dev = "cuda"
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(30, 128), (12, 200), (7, 256), (8, 512), (1, 512), (1, 128)]
    for (B, N) in shapes:
        R = B * N
        qkv = torch.randn(R, 3 * TD, device=dev)
        do = torch.randn(R, TD, device=dev)
        key_add = torch.zeros(B, N, device=dev)
        o = torch.empty(R, TD, device=dev)
        A = torch.empty(B, TH, N, N, device=dev)
        sc = 1.0 / math.sqrt(THD)
        L.call("fd_seq_attn_fwd", qkv, key_add, o, A, sc, B, N)
        dqkv = torch.empty(R, 3 * TD, device=dev)
        dA = torch.empty(B, TH, N, N, device=dev)

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

        A0 = A.clone()
        t_seq = timeit(lambda: (A.copy_(A0), seq()))       # (the row-softmax backward reads A; keep it intact per call)
        t_cp = timeit(lambda: A.copy_(A0))
        t_one = timeit(lambda: L.call("fd_seq_attn_bwd", qkv, A, do, o, dqkv, sc, B, N))
        print(f"B={B:3d} N={N:4d}: five launches {t_seq - t_cp:7.1f} us | one launch {t_one:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
