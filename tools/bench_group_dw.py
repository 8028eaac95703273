"""fd_group_dw against the split-K fd_gemm launches it replaces, on one trunk block's node-level weight gradients at the
training row count (B=30 x N=128 -> 3,840 rows).   python tools/bench_group_dw.py [rows] [blocks]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import ops, options  # noqa: E402
from se3_diffusion_amd.ops import mv  # noqa: E402

# (n_out, k_in, bias) of one trunk block: IPA projections + linear_out, skip, 2 transformer layers, post_tfmr, node transition,
# the per-residue halves of the edge transition
BLOCK = ([(6816, 256, True), (256, 2688, True), (64, 256, True)] + [(960, 320, True), (320, 320, True), (320, 320, True), (320, 320, True)] * 2
         + [(256, 320, True), (256, 256, True), (256, 256, True), (256, 256, True), (128, 256, True), (384, 128, False), (384, 128, True),
            (128, 128, False), (128, 128, True)])


def timeit(fn, n=20):
    for _ in range(3):
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
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
    dev = "cuda"
    ops.set_grad_stream(False)
    its = []
    for n, k, bias in BLOCK:
        its.append((torch.randn(rows, n, device=dev), torch.randn(rows, k, device=dev), torch.zeros(n, k, device=dev),
                    torch.zeros(n, device=dev) if bias else None, n, k))
    flops = sum(2.0 * rows * n * k for _, _, _, _, n, k in its)

    def grouped(blocks):
        with options.override(grouped_node_dw=True):
            for A, B, C, db, n, k in its:
                assert ops.queue_dw(mv(A), mv(B), mv(C), rows, n, k, db=db)
            ops.flush_dw(blocks=blocks)

    def unfused():
        for A, B, C, db, n, k in its:
            ops.linear_dw(mv(A), mv(B), mv(C), rows, n, k, db=db)

    for b in ([int(sys.argv[2])] if len(sys.argv) > 2 else [0, 256, 192, 128, 64]):
        ms = timeit(lambda: grouped(b))
        print(f"fd_group_dw blocks={b or 512:4d}  rows={rows}  {len(its)} items  {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
    ms = timeit(unfused)
    print(f"fd_gemm x {len(its)} (split-K)      rows={rows}  {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
    for t in its:
        t[2].zero_()
    grouped(0)
    a = [t[2].clone() for t in its]
    for t in its:
        t[2].zero_()
    unfused()
    torch.cuda.synchronize()
    print("max |grouped - gemm| / max:", max(float((x - t[2]).abs().max() / t[2].abs().max()) for x, t in zip(a, its)))


if __name__ == "__main__":
    main()
