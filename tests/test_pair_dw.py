"""Grouped pair-row weight gradients (csrc/fd_pair_dw.hip) against float64: the autograd of the three Linear layers of
EdgeTransition (model/ipa_pytorch.py:194-233), dW = dY^T X with the pair rows as the reduction index.

Tolerance: split-bf16 arithmetic is fp32-accurate -- 5e-6 of the tensor maximum (operands ~N(0,1); the sum runs over
`rows` products)."""
import os

import pytest
import torch

from se3_diffusion_amd import ops


def _case(dev, rows, seed, lda=384):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    wide = rn(rows, lda).to(dev)                      # d2 = 384 columns of a wider tensor when lda > 384 (row stride lda)
    t = dict(h1=rn(rows, 384), d1=rn(rows, 384), z=rn(rows, 128), h2=rn(rows, 384), dy=rn(rows, 128))
    t = {k: v.to(dev) for k, v in t.items()}
    t["d2"] = wide[:, 64:448] if lda > 384 else wide
    return t


def _run(dev, rows, seed=0, blocks=0, accumulate=False, lda=384):
    t = _case(dev, rows, seed, lda)
    mk = (lambda *s: torch.randn(*s, device=dev)) if accumulate else (lambda *s: torch.zeros(*s, device=dev))
    gW2, gb2, gW1, gWf = mk(384, 384), mk(384), mk(384, 384), mk(128, 384)
    ref0 = {k: v.double().cpu().clone() for k, v in dict(gW2=gW2, gb2=gb2, gW1=gW1, gWf=gWf).items()}
    d2 = t["d2"]
    lda = d2.stride(0)
    d2v = (d2, 0, lda)                                 # (a column slice is addressed by its own data pointer + the row stride)
    items = [dict(A=d2v, B=(t["h1"], 128 * j, 384), C=(gW2, 128 * j, 384), colsum=gb2 if j == 0 else None)
             for j in range(3)]
    items.append(dict(A=(t["d1"], 0, 384), B=(t["z"], 0, 128), C=(gW1, 0, 384)))
    # final layer: y = Wf (h2 + [z | e_i | e_j]) -> dWf = dy^T h2 with dy^T z added to its first 128 columns
    items.append(dict(A=(t["h2"], 0, 384), A_add=(t["z"], 0, 128), B=(t["dy"], 0, 128), C=(gWf, 0, 384), trans=True))
    ops.pair_dw(items, rows, blocks=blocks)
    d = {k: v.double().cpu() for k, v in t.items()}
    rW2 = ref0["gW2"] + d["d2"].T @ d["h1"]
    rb2 = ref0["gb2"] + d["d2"].sum(0)
    rW1 = ref0["gW1"].clone()
    rW1[:, :128] += d["d1"].T @ d["z"]
    rWf = ref0["gWf"] + d["dy"].T @ d["h2"]
    rWf[:, :128] += d["dy"].T @ d["z"]
    for got, ref, name in ((gW2, rW2, "W2"), (gb2, rb2, "b2"), (gW1, rW1, "W1"), (gWf, rWf, "Wf")):
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 5e-6, (name, err, rows, blocks)
    # untouched columns of gW1 (the e_i / e_j parts belong to other launches)
    assert torch.equal(gW1[:, 128:].double().cpu(), ref0["gW1"][:, 128:])


def _run_narrow(dev, rows, seed=0, blocks=0):
    """a_bands = 1 (A [rows,128]: a 128 x 128 tile) and b_cols = 120 (B [rows,120], C [128,120]) -- the edge embedder's layers
    (score_network.py:67-86): dW4 = dh3^T h2 (+ bias gradient), dW2 = dh2^T h1, dW0 = dh1^T x with x the 120 input features."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g).to(dev)
    dh3, dh2, dh1, h2, h1, x = rn(rows, 128), rn(rows, 128), rn(rows, 128), rn(rows, 128), rn(rows, 128), rn(rows, 120)
    gW4, gb4, gW2, gb2, gW0, gb0 = rn(128, 128), rn(128), rn(128, 128), rn(128), rn(128, 120), rn(128)
    ref = [t.double().cpu().clone() for t in (gW4, gb4, gW2, gb2, gW0, gb0)]
    items = [dict(A=(dh3, 0, 128), B=(h2, 0, 128), C=(gW4, 0, 128), colsum=gb4, a_bands=1),
             dict(A=(dh2, 0, 128), B=(h1, 0, 128), C=(gW2, 0, 128), colsum=gb2, a_bands=1),
             dict(A=(dh1, 0, 128), B=(x, 0, 120), C=(gW0, 0, 120), colsum=gb0, a_bands=1, b_cols=120)]
    ops.pair_dw(items, rows, blocks=blocks)
    d = lambda t: t.double().cpu()
    want = [ref[0] + d(dh3).T @ d(h2), ref[1] + d(dh3).sum(0), ref[2] + d(dh2).T @ d(h1), ref[3] + d(dh2).sum(0),
            ref[4] + d(dh1).T @ d(x), ref[5] + d(dh1).sum(0)]
    for got, w, name in zip((gW4, gb4, gW2, gb2, gW0, gb0), want, ("W4", "b4", "W2", "b2", "W0", "b0")):
        err = float((d(got) - w).abs().max() / w.abs().max())
        assert err < 5e-6, (name, err, rows, blocks)


def test_pair_dw_emu(use_emu):
    _run_narrow("cpu", rows=150, blocks=8)
    _run_narrow("cpu", rows=77, seed=1, blocks=16)
    _run("cpu", rows=150, blocks=8)                         # one group: 9 full stages + a ragged one
    _run("cpu", rows=200, seed=1, blocks=16, accumulate=True)   # three row ranges; C accumulates
    _run("cpu", rows=90, seed=2, blocks=8, lda=512)             # A = a column slice of a wider tensor


@pytest.mark.gpu
def test_pair_dw_gpu(hip_lib):
    _run("cuda", rows=150, blocks=8)
    _run("cuda", rows=12 * 12 * 3, seed=1)
    _run("cuda", rows=101 * 101, seed=2, accumulate=True)     # odd row count: ragged last stage
    _run("cuda", rows=2 * 128 * 128, seed=3)
    _run("cuda", rows=30 * 128 * 128, seed=4)                 # the training shape
    _run("cuda", rows=30 * 128 * 128, seed=5, blocks=160)     # ... on 160 CUs, as the training step launches it
    _run("cuda", rows=5000, seed=6, lda=512)                  # A = a column slice of a wider tensor
    _run_narrow("cuda", rows=150, blocks=8)
    _run_narrow("cuda", rows=101 * 101, seed=1)
    _run_narrow("cuda", rows=30 * 128 * 128, seed=2)
