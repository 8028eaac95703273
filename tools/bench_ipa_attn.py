"""Microbenchmark of the IPA per-row attention kernels (logits + softmax [+ o_pair] and their backward) at B x N."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd.ops import lib  # noqa: E402

H, PQ, ZB, LDF = 8, 8, 40, 2688


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = "cuda"
    R, P = B * N, B * N * N
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    zb, qp, kp, hw, mask = rn(P, ZB), rn(R, H, PQ * 3), rn(R, H, PQ * 3), rn(H), torch.ones(R, device=dev)
    S0, feats, dfeats, dA0 = rn(B, H, N, N), torch.zeros(R, LDF, device=dev), rn(R, LDF), rn(B, H, N, N)
    S = S0.clone()
    L = lib()
    t1 = timeit(lambda: L.call("fd_ipa_softmax_fwd", S, zb, qp, kp, hw, mask, B, N))
    t2 = timeit(lambda: L.call("fd_ipa_opair_fwd", S, zb, feats, B, N))
    kpT = kp.reshape(B, N, H, PQ * 3).permute(0, 2, 3, 1).contiguous()
    t3 = timeit(lambda: L.call("fd_ipa_attn_fwd", S, zb, qp, kp, None, hw, mask, feats, B, N))
    t3s = timeit(lambda: L.call("fd_ipa_attn_fwd", S, zb, qp, kp, kpT, hw, mask, feats, B, N))
    L.call("fd_ipa_softmax_fwd", S, zb, qp, kp, hw, mask, B, N)
    dA, dzb = dA0.clone(), torch.empty(P, ZB, device=dev)
    dqp, dkp, dhw, part = torch.empty(R, H, PQ * 3, device=dev), torch.empty(R, H, PQ * 3, device=dev), torch.zeros(H, device=dev), torch.empty(R, H, device=dev)
    t4 = timeit(lambda: L.call("fd_ipa_attn_bwd", S, dA, zb, dfeats, qp, kp, None, hw, dzb, dqp, dkp, dhw, part, B, N))
    t4s = timeit(lambda: L.call("fd_ipa_attn_bwd", S, dA, zb, dfeats, qp, kp, kpT, hw, dzb, dqp, dkp, dhw, part, B, N))
    print(f"B={B} N={N}: softmax_fwd {t1:.1f} us | opair_fwd {t2:.1f} us | attn_fwd (both) {t3:.1f} us, key points from the "
          f"[B,8,24,N] copy {t3s:.1f} us | attn_bwd (+kpts, colsum) {t4:.1f} us, with the copy {t4s:.1f} us")


if __name__ == "__main__":
    main()
