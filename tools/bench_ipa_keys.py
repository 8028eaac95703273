"""Key side of the IPA attention backward: fd_ipa_flash_bwd_keys (one launch) against the launches it replaces (A^T dO, A^T dO_pt,
dL^T Q as batched fd_gemm + fd_ipa_kpts_bwd).   python tools/bench_ipa_keys.py [BxN ...]   (GPU box)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import hip  # noqa: E402
from tools.bench_node_gemm import timeit  # noqa: E402

H, C, PQ, PV, LDP, LDF = 8, 256, 8, 12, 6816, 2688


def main():
    L = hip.get_lib()
    dev = "cuda"
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(30, 128), (12, 200), (7, 256), (8, 512), (1, 512)]
    for (B, N) in shapes:
        R = B * N
        A = torch.softmax(torch.randn(B, H, N, N, device=dev), -1)
        dL = torch.randn(B, H, N, N, device=dev) * 0.1
        proj, dfeats, doptg = torch.randn(R, LDP, device=dev), torch.randn(R, LDF, device=dev), torch.randn(R, H, PV * 3, device=dev)
        qp, kp, hw = torch.randn(R, H, PQ * 3, device=dev), torch.randn(R, H, PQ * 3, device=dev), torch.randn(H, device=dev)
        dproj, dvp, dkp = torch.empty(R, LDP, device=dev), torch.empty(R, H, PV * 3, device=dev), torch.empty(R, H, PQ * 3, device=dev)
        sc = math.sqrt(1.0 / (3 * C))

        def seq():
            L.gemm(A, dfeats, dproj, N, C, N, (1, N), (LDF, 1), LDP, c_off=2048 + C, batch=B * H, bdiv=H,
                   a_bs=(H * N * N, N * N), b_bs=(N * LDF, C), c_bs=(N * LDP, 2 * C))
            L.gemm(A, doptg, dvp, N, PV * 3, N, (1, N), (H * PV * 3, 1), H * PV * 3, batch=B * H, bdiv=H,
                   a_bs=(H * N * N, N * N), b_bs=(N * H * PV * 3, PV * 3), c_bs=(N * H * PV * 3, PV * 3))
            L.gemm(dL, proj, dproj, N, C, N, (1, N), (LDP, 1), LDP, c_off=2048, batch=B * H, bdiv=H,
                   a_bs=(H * N * N, N * N), b_bs=(N * LDP, C), c_bs=(N * LDP, 2 * C), alpha=sc)
            L.call("fd_ipa_kpts_bwd", dL, qp, kp, hw, dkp, B, N)

        line = [f"B={B:3d} N={N:4d}: launch sequence {timeit(seq):7.1f} us | one launch"]
        for hpb in (1, 2, 4, 8):
            t = timeit(lambda: L.call("fd_ipa_flash_bwd_keys", A, dL, proj, dfeats, doptg, qp, kp, hw, dproj, dvp, dkp, B, N, hpb))
            line.append(f"hpb {hpb}: {t:7.1f}")
        print(" ".join(line), flush=True)


if __name__ == "__main__":
    main()
