"""Node-level GEMM shapes of the training step (M = B*N = 3840 rows) on every tile that accepts them, forward layout
(y = x W^T, weights k-contiguous) and activation-gradient layout (dx = dy W, weights row-contiguous); tiles 12 / 13 / 14 read
the weights as pre-split bf16 planes (fd_split_planes).   python tools/bench_node_gemm.py [rows ...]   (GPU box)"""
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
    rows = [int(a) for a in sys.argv[1:]] or [3840]
    # (N_out, K_in) of the nn.Linear; the dX product of the same layer is [M, N_out] x [N_out, K_in]
    layers = [(320, 320), (256, 256), (960, 320), (320, 1280), (256, 2688), (6816, 256), (128, 256), (64, 256)]
    tiles = (0, 2, 10, 4, 12, 13, 14)
    for M in rows:
        for mode in ("fwd", "dx"):
            print(f"M = {M}, {mode}: us per launch (back to back) | TFLOP/s of the best;  tiles {tiles}; 12-14 = pre-split weight planes"
                  + (", ks = split-K of the long reductions" if mode == "dx" else ""))
            for (No, Ki) in layers:
                W = torch.randn(No, Ki, device=dev)
                pl = torch.empty((3, W.numel()), dtype=torch.int16, device=dev)
                lib.call("fd_split_planes", W, W.numel(), pl)
                planes = (pl.data_ptr(), W.numel())
                b = torch.randn(No, device=dev)
                if mode == "fwd":
                    A = torch.randn(M, Ki, device=dev); C = torch.empty(M, No, device=dev)
                    Ng, Kg, b_str, kw = No, Ki, (1, Ki), dict(bias=b)
                else:
                    A = torch.randn(M, No, device=dev); C = torch.zeros(M, Ki, device=dev)
                    Ng, Kg, b_str, kw = Ki, No, (Ki, 1), {}
                line = [f"  N={Ng:5d} K={Kg:5d}"]
                best = 1e9
                for tile in tiles:
                    splits = (1,) if (mode == "fwd" and Kg < 2048) or Kg < 960 else (1, 4, 8)
                    for ks in splits:
                        if ks > 1 and tile not in (10, 12, 13, 14):
                            continue
                        k2 = dict(kw) if ks == 1 else {}
                        try:
                            t = timeit(lambda: lib.gemm(A, W, C, M, Ng, Kg, (Kg, 1), b_str, Ng, tile=tile, ksplit=ks,
                                                        b_planes=planes if tile in (0, 12, 13, 14) else None, **k2))
                            line.append(f"{tile}{'/ks' + str(ks) if ks > 1 else ''}: {t:6.1f}")
                            best = min(best, t)
                        except Exception:  # noqa: BLE001
                            line.append(f"{tile}:    n/a")
                print(" | ".join(line) + f" | {2.0 * M * Ng * Kg / best / 1e6:6.1f}", flush=True)


if __name__ == "__main__":
    main()
