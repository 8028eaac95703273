"""Micro-benchmark fd_gemm on the shapes that dominate the FrameDiff step (MI355X).
   python tools/bench_gemm.py [--only substr] [--iters n] [--warm n]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import hip  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


P = 30 * 128 * 128
SHAPES = [
    # name, M, N, K, a_kc, b_kc, tile, ksplit
    ("edge_fwd_W2 NT", P, 384, 384, True, True, 1, 1),
    ("edge_fwd_W1z NT", P, 384, 128, True, True, 1, 1),
    ("edge_fwd_Wf NT", P, 128, 384, True, True, 1, 1),
    ("edge_bwd_dX NN", P, 384, 384, True, False, 1, 1),
    ("edge_bwd_dW TN", 384, 384, P, False, False, 1, 256),
    ("edge_bwd_dW TN t2", 384, 384, P, False, False, 2, 128),
    ("ipa_proj NT", 3840, 6816, 256, True, True, 1, 1),
    ("ipa_proj NT t2", 3840, 6816, 256, True, True, 2, 1),
    ("ipa_out NT", 3840, 256, 2688, True, True, 2, 1),
    ("z_to_40 NT", P, 40, 128, True, True, 3, 1),
    ("z_to_40 NT t2", P, 40, 128, True, True, 2, 1),
    ("z_to_40 NT t10", P, 40, 128, True, True, 10, 1),
    ("dz_from_40 NN beta t10", P, 128, 40, True, False, 10, 1),
    ("z_to_40 NT t4", P, 40, 128, True, True, 4, 1),
    ("z_to_40 NT t1", P, 40, 128, True, True, 1, 1),
    ("dz_from_40 NN beta t1", P, 128, 40, True, False, 1, 1),
    ("dz_from_40 NN beta t2", P, 128, 40, True, False, 2, 1),
    ("sample N=128 proj", 128, 6816, 256, True, True, 2, 1),
    ("sample edge W2 N=128", 16384, 384, 384, True, True, 1, 1),
    ("sample edge W2 N=128 t2", 16384, 384, 384, True, True, 2, 1),
    ("square 4096 NT", 4096, 4096, 4096, True, True, 1, 1),
    ("probe shortK bigN", 32768, 4096, 384, True, True, 1, 1),
    ("probe longK N384", 122880, 384, 1536, True, True, 1, 1),
    ("probe K384 N384 M64k", 65536, 384, 384, True, True, 1, 1),
    # split-bf16, two blocks per CU (tile 6)
    ("y6 edge_fwd_W2 NT", P, 384, 384, True, True, 6, 1),
    ("y6 edge_fwd_W1z NT", P, 384, 128, True, True, 6, 1),
    ("y6 edge_fwd_Wf NT", P, 128, 384, True, True, 6, 1),
    ("y6 edge_fwd_Wfz NT", P, 128, 128, True, True, 6, 1),
    ("y6 edge_bwd_dX NN", P, 384, 384, True, False, 6, 1),
    ("y6 edge_bwd_dX NN N128", P, 128, 384, True, False, 6, 1),
    ("y6 edge_bwd_dW TN", 384, 384, P, False, False, 6, 128),
    ("y6 dW 384x128", 384, 128, P, False, False, 6, 256),
    ("y6 dW 128x384", 128, 384, P, False, False, 6, 256),
    ("y6 dW 128x128", 128, 128, P, False, False, 6, 768),
    ("y6 sample edge W2 N=128", 16384, 384, 384, True, True, 6, 1),
    ("y6 square 4096 NT", 4096, 4096, 4096, True, True, 6, 1),
    # split-bf16 (tile 4) on the same pair-level shapes
    ("x3 edge_fwd_W2 NT", P, 384, 384, True, True, 4, 1),
    ("x3 edge_fwd_W1z NT", P, 384, 128, True, True, 4, 1),
    ("x3 edge_fwd_Wf NT", P, 128, 384, True, True, 4, 1),
    ("x3 edge_fwd_Wfz NT", P, 128, 128, True, True, 4, 1),
    ("x3 edge_bwd_dX NN", P, 384, 384, True, False, 4, 1),
    ("x3 edge_bwd_dX NN N128", P, 128, 384, True, False, 4, 1),
    ("x3 edge_bwd_dW TN", 384, 384, P, False, False, 4, 384),
    ("x3 edge_bwd_dW TN 128", 128, 384, P, False, False, 4, 768),
    ("x3 dWks 384x128 ks96", 384, 128, P, False, False, 4, 96),
    ("x3 dWks 384x128 ks192", 384, 128, P, False, False, 4, 192),
    ("x3 dWks 384x128 ks384", 384, 128, P, False, False, 4, 384),
    ("t2 dWks 384x128 ks192", 384, 128, P, False, False, 2, 192),
    ("node dWks 15", 320, 320, 3840, False, False, 2, 15),
    ("node dWks 30", 320, 320, 3840, False, False, 2, 30),
    ("node dWks 60", 320, 320, 3840, False, False, 2, 60),
    ("node dWks 256x2688 13", 256, 2688, 3840, False, False, 2, 13),
    ("node dWks 256x2688 4", 256, 2688, 3840, False, False, 2, 4),
    ("node dWks 256x256 15", 256, 256, 3840, False, False, 2, 15),
    ("node dWks 256x256 60", 256, 256, 3840, False, False, 2, 60),
    ("lat t2 128x320x320", 128, 320, 320, True, True, 2, 1),
    ("lat t5 128x320x320", 128, 320, 320, True, True, 5, 1),
    ("lat t2 128x6816x256", 128, 6816, 256, True, True, 2, 1),
    ("lat t5 128x6816x256", 128, 6816, 256, True, True, 5, 1),
    ("lat t2 128x256x2688", 128, 256, 2688, True, True, 2, 1),
    ("lat t5 128x256x2688", 128, 256, 2688, True, True, 5, 1),
    ("lat t2 1024x320x320", 1024, 320, 320, True, True, 2, 1),
    ("lat t5 1024x320x320", 1024, 320, 320, True, True, 5, 1),
    ("lat t2 1024x960x320", 1024, 960, 320, True, True, 2, 1),
    ("lat t5 1024x960x320", 1024, 960, 320, True, True, 5, 1),
    ("kk K=32", 3840, 320, 32, True, True, 2, 1),
    ("kk K=64", 3840, 320, 64, True, True, 2, 1),
    ("kk K=160", 3840, 320, 160, True, True, 2, 1),
    ("kk K=320", 3840, 320, 320, True, True, 2, 1),
    ("kk K=640", 3840, 320, 640, True, True, 2, 1),
    ("kk K=1280", 3840, 320, 1280, True, True, 2, 1),
    ("kk s64 K=32", 3840, 320, 32, True, True, 10, 1),
    ("kk s64 K=64", 3840, 320, 64, True, True, 10, 1),
    ("kk s64 K=160", 3840, 320, 160, True, True, 10, 1),
    ("kk s64 K=320", 3840, 320, 320, True, True, 10, 1),
    ("kk s64 K=640", 3840, 320, 640, True, True, 10, 1),
    ("kk s64 K=1280", 3840, 320, 1280, True, True, 10, 1),
    ("kk s64 NN K=320", 3840, 320, 320, True, False, 10, 1),
    ("kk s64 TN ks15", 320, 320, 3840, False, False, 10, 15),
    ("kk t2 TN ks15", 320, 320, 3840, False, False, 2, 15),
    ("kk s64 TN ks30", 320, 320, 3840, False, False, 10, 30),
    ("n5 NT 3840x320x320 t5", 3840, 320, 320, True, True, 5, 1),
    ("n5 NT 3840x320x320 t2", 3840, 320, 320, True, True, 2, 1),
    ("n5 NN 3840x320x320 t5", 3840, 320, 320, True, False, 5, 1),
    ("n5 NN 3840x320x320 t2", 3840, 320, 320, True, False, 2, 1),
    ("n5 NT 3840x960x320 t5", 3840, 960, 320, True, True, 5, 1),
    ("n5 NT 3840x960x320 t2", 3840, 960, 320, True, True, 2, 1),
    ("n5 NT 3840x256x2688 t5", 3840, 256, 2688, True, True, 5, 1),
    ("n5 NT 3840x256x2688 t2", 3840, 256, 2688, True, True, 2, 1),
    ("n5 NT 3840x256x256 t5", 3840, 256, 256, True, True, 5, 1),
    ("n5 NT 3840x256x256 t2", 3840, 256, 256, True, True, 2, 1),
    ("n5 NN 3840x2688x256 t5", 3840, 2688, 256, True, False, 5, 1),
    ("n5 NN 3840x2688x256 t2", 3840, 2688, 256, True, False, 2, 1),
    ("node NT 3840x320x320", 3840, 320, 320, True, True, 2, 1),
    ("node NT 3840x320x320 t3", 3840, 320, 320, True, True, 3, 1),
    ("node NN 3840x320x320", 3840, 320, 320, True, False, 2, 1),
    ("x3 dWks 64", 384, 384, P, False, False, 4, 64),
    ("x3 dWks 96", 384, 384, P, False, False, 4, 96),
    ("x3 dWks 128", 384, 384, P, False, False, 4, 128),
    ("x3 dWks 192", 384, 384, P, False, False, 4, 192),
    ("t2 dWks 48 (128x384)", 128, 384, P, False, False, 2, 48),
    ("t2 dWks 96 (128x384)", 128, 384, P, False, False, 2, 96),
    ("t2 dWks 192 (128x384)", 128, 384, P, False, False, 2, 192),
    ("t2 dWks 144 (128x128)", 128, 128, P, False, False, 2, 144),
    ("t2 dWks 288 (128x128)", 128, 128, P, False, False, 2, 288),
    ("t2 dWks 576 (128x128)", 128, 128, P, False, False, 2, 576),
    ("ov K=64", P, 384, 64, True, True, 4, 1),
    ("ov K=128", P, 384, 128, True, True, 4, 1),
    ("ov K=256", P, 384, 256, True, True, 4, 1),
    ("ov K=384", P, 384, 384, True, True, 4, 1),
    ("ov K=768", P, 384, 768, True, True, 4, 1),
    ("ov K=1536", P, 384, 1536, True, True, 4, 1),
    ("x3 sample edge W2 N=128", 16384, 384, 384, True, True, 4, 1),
    ("x3 sample edge W2 N=256", 65536, 384, 384, True, True, 4, 1),
    ("x3 square 4096 NT", 4096, 4096, 4096, True, True, 4, 1),
    ("x3 probe longK N384", 122880, 384, 1536, True, True, 4, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--mtiles", type=int, default=0)
    ap.add_argument("--fill", default="randn", help="randn | ones | zeros (data-dependent power check)")
    a = ap.parse_args()
    lib = hip.get_lib()
    dev = "cuda"
    out = []
    for name, M, N, K, akc, bkc, tile, ks in SHAPES:
        if a.only and a.only not in name:
            continue
        A = torch.randn(M, K, device=dev) if akc else torch.randn(K, M, device=dev)
        B = torch.randn(N, K, device=dev) if bkc else torch.randn(K, N, device=dev)
        if a.fill != "randn":
            A.fill_(1.0 if a.fill == "ones" else 0.0)
            B.fill_(1.0 if a.fill == "ones" else 0.0)
        C = torch.zeros(M, N, device=dev)
        a_str = (K, 1) if akc else (1, M)
        b_str = (1, K) if bkc else (N, 1)
        fn = lambda: lib.gemm(A, B, C, M, N, K, a_str, b_str, N, tile=tile, ksplit=ks, mtiles=a.mtiles)  # noqa: E731
        ms = timeit(fn, a.iters, a.warm)
        tf = 2.0 * M * N * K / ms / 1e9
        out.append(dict(name=name, M=M, N=N, K=K, tile=tile, ksplit=ks, ms=round(ms, 4), tflops=round(tf, 2)))
        print(f"{name:28s} M={M:7d} N={N:5d} K={K:7d} tile={tile} ks={ks:3d}  {ms:8.3f} ms  {tf:7.2f} TF/s", flush=True)
        del A, B, C
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_gemm.json", "w"), indent=1)


if __name__ == "__main__":
    main()
