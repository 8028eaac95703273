"""Does the ROW STRIDE of the operands set fd_pair_dw's stage time?  Eight 384 x 128 items over the same rows, A either eight
contiguous [rows, 384] tensors or eight 384-column slabs of one [rows, 6816] tensor (what a unit of fd_group_dw reads of the IPA
projections' dY), B [rows, 128] contiguous or slabs of [rows, 256].   python tools/probes/pair_dw_stride.py [rows]   (GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from se3_diffusion_amd import ops  # noqa: E402


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
    dev = "cuda"
    wide = torch.randn(rows, 6816, device=dev)
    xw = torch.randn(rows, 256, device=dev)
    As = [torch.randn(rows, 384, device=dev) for _ in range(8)]
    Bs = [torch.randn(rows, 128, device=dev) for _ in range(8)]
    C = [torch.zeros(384, 128, device=dev) for _ in range(8)]
    cases = {
        "A contiguous [rows,384], B contiguous [rows,128]": [dict(A=(As[i], 0, 384), B=(Bs[i], 0, 128), C=(C[i], 0, 128)) for i in range(8)],
        "A slabs of [rows,6816],  B contiguous [rows,128]": [dict(A=(wide, 384 * i, 6816), B=(Bs[i], 0, 128), C=(C[i], 0, 128)) for i in range(8)],
        "A slabs of [rows,6816],  B slabs of [rows,256]   ": [dict(A=(wide, 384 * (i // 2), 6816), B=(xw, 128 * (i % 2), 256), C=(C[i], 0, 128)) for i in range(8)],
        "A contiguous,            B slabs of [rows,256]   ": [dict(A=(As[i // 2], 0, 384), B=(xw, 128 * (i % 2), 256), C=(C[i], 0, 128)) for i in range(8)],
    }
    nst = rows / 16
    for name, items in cases.items():
        for blocks in (0, 128):
            ms = timeit(lambda: ops.pair_dw(items, rows, blocks=blocks))
            nb = blocks or 256
            print(f"{name}  blocks={nb:3d}: {ms:7.3f} ms   {2.0 * rows * 384 * 128 * 8 / ms / 1e9:6.1f} TFLOP/s   {ms * 1e3 / (nst * 8 / nb):5.2f} us per stage and block", flush=True)


if __name__ == "__main__":
    main()
