"""fd_group_dw on chosen item lists at 3,840 rows with the descriptor built ONCE and the C entry called back to back (no Python between
the launches: the time is the kernel's): what a stage of each unit shape costs.
   python tools/bench_group_dw_items.py [rows]   (GPU box; FD_GROUP_DW_V1 / FD_GROUP_DW_LOCKSTEP select the kernel form)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import hip  # noqa: E402

BLOCK = ([(6816, 256), (256, 2688), (64, 256)] + [(960, 320), (320, 320), (320, 320), (320, 320)] * 2
         + [(256, 320), (256, 256), (256, 256), (256, 256), (128, 256), (384, 128), (384, 128), (128, 128), (128, 128)])
CASES = [("one trunk block (20 items)", BLOCK), ("IPA projections 6816 x 256", [(6816, 256)]), ("2 x 6816 x 256", [(6816, 256)] * 2),
         ("linear_out 256 x 2688", [(256, 2688)]), ("4 x 256 x 2688", [(256, 2688)] * 4), ("2 x in_proj 960 x 320", [(960, 320)] * 2),
         ("8 x 960 x 320", [(960, 320)] * 8), ("6 x 320 x 320", [(320, 320)] * 6), ("24 x 320 x 320", [(320, 320)] * 24),
         ("20 x 256 x 256", [(256, 256)] * 20), ("32 x 128 x 128", [(128, 128)] * 32), ("32 x 384 x 128", [(384, 128)] * 32)]


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
    dev = "cuda"
    L = hip.get_lib()
    stream = torch.cuda.current_stream().cuda_stream
    for name, shapes in CASES:
        d = hip.FdGroupDwDesc()
        keep = []
        for t, (n, k) in enumerate(shapes):
            A, B, C, db = torch.randn(rows, n, device=dev), torch.randn(rows, k, device=dev), torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)
            keep += [A, B, C, db]
            e = d.item[t]
            e.A, e.B, e.C, e.a_colsum = hip._ptr(A), hip._ptr(B), hip._ptr(C), hip._ptr(db)
            e.lda, e.ldb, e.ldc, e.n_out, e.k_in = n, k, k, n, k
        d.nitems, d.rows, d.blocks = len(shapes), rows, 0
        ref = hip.ctypes.byref(d)
        for _ in range(3):
            L._check(L.cdll.fd_group_dw(ref, stream), "fd_group_dw")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            L.cdll.fd_group_dw(ref, stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        fl = sum(2.0 * rows * n * k for n, k in shapes)
        print(f"{name:32s} {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
