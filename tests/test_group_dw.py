"""fd_group_dw (csrc/fd_group_dw.hip): the node-level weight / bias gradients of a trunk block in one launch, against
float64 -- dW_t += dY_t^T X_t, db_t += sum_r dY_t (autograd of nn.Linear, model/ipa_pytorch.py:169-191,236-301,584-595).

Shapes cover what the backward pass queues (320 x 320, 960 x 320, 256 x 2688, 64 x 256, 6816 x 256 as a slice) plus the edge
cases of the tiling: column tails (n_out, k_in not multiples of 128, down to 4), operands that are column slices of wider
tensors, C as a column slice of a wider gradient, accumulation into non-zero C, ragged last stage, fewer stages than row
ranges, more units than blocks.  Tolerance 5e-6 of the result's maximum: split-bf16 products are fp32-accurate."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import ops, options  # noqa: E402
from se3_diffusion_amd.ops import mv  # noqa: E402

SHAPES_SMALL = [(320, 320, True), (64, 256, True), (200, 132, False), (4, 8, True), (136, 4, True), (960, 320, False)]
SHAPES_FULL = [(6816, 256, True), (256, 2688, True), (960, 320, True), (320, 320, True), (256, 320, False), (64, 256, True),
               (384, 128, True), (128, 128, False)]


def _run(dev, rows, shapes, seed=0, blocks=0, slices=True):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    items, refs = [], []
    for i, (n, k, bias) in enumerate(shapes):
        pad_a, pad_b, pad_c = (8, 8, 12) if (slices and i % 2 == 0) else (0, 0, 0)
        A = rn(rows, n + pad_a)                    # dY as a column slice [:, pad_a/2 : pad_a/2 + n]
        Bm = rn(rows, k + pad_b)
        C = rn(n, k + pad_c)                       # accumulate into a non-zero gradient, column slice of a wider one
        db = rn(n) if bias else None
        oa, ob, oc = pad_a // 2, pad_b // 2, pad_c // 3         # (4, 4, 4): float4-addressable operand slices
        refs.append((C.double().cpu().clone(), None if db is None else db.double().cpu().clone(),
                     A[:, oa:oa + n].double().cpu(), Bm[:, ob:ob + k].double().cpu(), oc))
        items.append((A, oa, Bm, ob, C, oc, db, n, k))
    with options.override(grouped_node_dw=True):
        for A, oa, Bm, ob, C, oc, db, n, k in items:
            assert ops.queue_dw((A, oa, A.shape[1]), (Bm, ob, Bm.shape[1]), (C, oc, C.shape[1]), rows, n, k, db=db)
        ops.flush_dw(blocks=blocks)
        ops.join_grad_stream()
    if dev != "cpu":
        torch.cuda.synchronize()
    for (A, oa, Bm, ob, C, oc, db, n, k), (C0, db0, Ad, Bd, oc_) in zip(items, refs):
        want = C0.clone()
        want[:, oc:oc + k] += Ad.T @ Bd
        err = float((C.double().cpu() - want).abs().max() / want.abs().max())
        assert err < 5e-6, ("W", n, k, rows, err)
        if db is not None:
            wb = db0 + Ad.sum(0)
            errb = float((db.double().cpu() - wb).abs().max() / wb.abs().max())
            assert errb < 5e-6, ("b", n, k, rows, errb)


def test_group_dw_emu(use_emu):
    _run("cpu", rows=150, shapes=SHAPES_SMALL, blocks=3)            # many units per block, ragged last stage
    _run("cpu", rows=16, shapes=SHAPES_SMALL[:3], seed=1, blocks=64)    # one stage per tile
    _run("cpu", rows=7, shapes=SHAPES_SMALL[1:4], seed=2)              # less than a stage
    _run("cpu", rows=300, shapes=[(132, 260, True)], seed=3, blocks=2, slices=False)   # row ranges (nsplit > 1)


def test_queue_rejects_what_the_kernel_cannot_address(use_emu):
    A, B, C = torch.randn(8, 6), torch.randn(8, 256), torch.zeros(6, 256)
    with options.override(grouped_node_dw=True):
        assert not ops.queue_dw(mv(A), mv(B), mv(C), 8, 6, 256)            # bb_update: n_out = 6
        assert not ops.queue_dw(mv(torch.randn(8, 256)), mv(torch.randn(8, 65)), mv(torch.zeros(256, 65)), 8, 256, 65)
    with options.override(grouped_node_dw=False):
        assert not ops.queue_dw(mv(B), mv(B), mv(torch.zeros(256, 256)), 8, 256, 256)
    assert not ops._DWQ["items"]


@pytest.mark.gpu
def test_group_dw_gpu(hip_lib):
    _run("cuda", rows=150, shapes=SHAPES_SMALL, blocks=3)
    _run("cuda", rows=16, shapes=SHAPES_SMALL[:3], seed=1, blocks=64)
    _run("cuda", rows=3840, shapes=SHAPES_FULL, seed=2)               # the training step's row count, a block's big items
    _run("cuda", rows=3840, shapes=SHAPES_FULL, seed=3, blocks=128)
    _run("cuda", rows=101, shapes=SHAPES_FULL[2:], seed=4)
    _run("cuda", rows=1024, shapes=SHAPES_FULL[1:5], seed=5, slices=False)
