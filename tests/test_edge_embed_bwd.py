"""fd_edge_embed_bwd (csrc/fd_edge_embed_bwd.hip): LayerNorm backward + the two gated dX products of the edge embedder in one
launch, against a float64 restatement of autograd through  z = rowscale * LayerNorm(W4 relu(W2 relu(h1pre)) ...)
(model/score_network.py:67-86,194-195).  Bounds: 2e-5 of each tensor's maximum (split-bf16 products are fp32-accurate)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import ops  # noqa: E402


def _run(dev, rows, seed=0, blocks=0, mask=True):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    C = 128
    W2, W4 = rn(C, C) / 8, rn(C, C) / 8
    gamma = 1 + 0.3 * rn(C)
    h1 = torch.relu(rn(rows, C)); h2 = torch.relu(rn(rows, C)); h3 = 2 * rn(rows, C) + 0.5
    dy = rn(rows, C)
    rowscale = (torch.rand(rows, generator=g) > 0.2).float() if mask else None
    mean = h3.mean(-1); var = h3.var(-1, unbiased=False); rstd = 1 / torch.sqrt(var + 1e-5)
    # float64 reference
    D = lambda t: t.double()
    xh = (D(h3) - D(mean)[:, None]) * D(rstd)[:, None]
    gy = D(dy) * (D(rowscale)[:, None] if mask else 1.0)
    t = gy * D(gamma)
    dh3 = D(rstd)[:, None] * (t - t.mean(-1, keepdim=True) - xh * (t * xh).mean(-1, keepdim=True))
    dh2 = (dh3 @ D(W4)) * (D(h2) > 0)
    dh1 = (dh2 @ D(W2)) * (D(h1) > 0)
    dgam0, dbet0 = rn(C), rn(C)
    want = dict(dh3=dh3, dh2=dh2, dh1=dh1, dgamma=D(dgam0) + (gy * xh).sum(0), dbeta=D(dbet0) + gy.sum(0))
    to = lambda x: None if x is None else x.to(dev).contiguous()
    o3, o2, o1 = (torch.full((rows, C), float("nan"), device=dev) for _ in range(3))
    dgam, dbet = to(dgam0.clone()), to(dbet0.clone())
    img = ops.edge_embed_bwd_pack(to(W2), to(W4))
    ops.edge_embed_bwd(to(dy), to(h3), to(mean), to(rstd), to(gamma), to(rowscale), to(h2), to(h1), img, o3, o2, o1, dgam, dbet,
                       rows, blocks=blocks)
    got = dict(dh3=o3, dh2=o2, dh1=o1, dgamma=dgam, dbeta=dbet)
    for k, w in want.items():
        err = float((got[k].double().cpu() - w).abs().max() / w.abs().max())
        assert err < 2e-5, (k, err, rows, blocks)
    # the same with the ReLU gates as packed sign bits (fd_edge_embed's mask outputs: bit 4 nb + e of word (row, g) <-> unit
    # 16 nb + 4 g + e) instead of reads of h2 / h1: bit-identical
    unit = torch.arange(C)
    nb, gg, ee = unit // 16, (unit % 16) // 4, unit % 4
    def pack(h):
        w = torch.zeros(rows, 4, dtype=torch.int64)
        for u in range(C):
            w[:, gg[u]] |= (h[:, u] > 0).long() << int(4 * nb[u] + ee[u])
        return (w & 0xFFFFFFFF).to(torch.int64).apply_(lambda v: v - (1 << 32) if v >= (1 << 31) else v).to(torch.int32).to(dev)
    p3, p2, p1 = (torch.full((rows, C), float("nan"), device=dev) for _ in range(3))
    dg2, db2 = to(dgam0.clone()), to(dbet0.clone())
    ops.edge_embed_bwd(to(dy), to(h3), to(mean), to(rstd), to(gamma), to(rowscale), None, None, img, p3, p2, p1, dg2, db2, rows,
                       blocks=blocks, gmask2=pack(h2), gmask1=pack(h1))
    assert torch.equal(p3, o3) and torch.equal(p2, o2) and torch.equal(p1, o1)


def test_edge_embed_bwd_emu(use_emu):
    _run("cpu", rows=150, blocks=2)                  # persistent blocks walk two tiles, ragged last tile
    _run("cpu", rows=300, seed=3, blocks=1)          # five tiles on one block: the dynamic tile hand-out
    _run("cpu", rows=64, seed=1, mask=False)
    _run("cpu", rows=5, seed=2, blocks=8)            # less than one wave's rows; idle blocks


@pytest.mark.gpu
def test_edge_embed_bwd_gpu(hip_lib):
    _run("cuda", rows=150, blocks=2)
    _run("cuda", rows=101 * 101, seed=1)
    _run("cuda", rows=2 * 128 * 128, seed=2, mask=False)
    _run("cuda", rows=30 * 128 * 128, seed=3)        # the training shape
