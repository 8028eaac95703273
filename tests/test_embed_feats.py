"""Kernel-level checks of the embedder's per-residue features and of the fused edge embedder's strided node terms:
fd_node_feats_ld (the [R, 65] features of score_network.py:14-47,97-126 at a row stride >= 65, zero columns behind them) against
fd_node_feats and a float64 restatement; fd_edge_embed reading p | q as the column halves of one [R, 256] buffer
(FdEdgeEmbedDesc.ld_pq) against two dense [R, 128] tensors, bit for bit."""
import math

import numpy as np
import pytest
import torch

from se3_diffusion_amd import ops
from se3_diffusion_amd.ops import lib


def _node_feats(dev, B, N, seed):
    g = torch.Generator().manual_seed(seed)
    seq = torch.arange(N).repeat(B, 1).reshape(-1).to(torch.int64).to(dev)
    t = torch.rand(B, generator=g)
    tscaled = (t * 10000).float().to(dev)
    fixed = (torch.rand(B * N, generator=g) > 0.7).float().to(dev)
    tfreq, idenom, _, _ = ops.feature_tables(torch.device(dev))
    a = torch.full((B * N, 65), float("nan"), device=dev)
    lib().call("fd_node_feats", seq, tscaled, fixed, tfreq, idenom, a, B, N)
    b = torch.full((B * N, 72), float("nan"), device=dev)
    lib().call("fd_node_feats_ld", seq, tscaled, fixed, tfreq, idenom, b, 72, B, N)
    assert torch.equal(b[:, :65], a) and float(b[:, 65:].abs().max()) == 0.0
    # float64: [sin | cos](t 1e4 freq) (32), fixed (1), [sin | cos](idx pi / denom) (32)
    tf, idn = tfreq.double().cpu().numpy(), idenom.double().cpu().numpy()
    ts = tscaled.double().cpu().numpy().repeat(N)
    arg = ts[:, None] * tf[None, :]
    idx = seq.double().cpu().numpy()
    ia = idx[:, None] * math.pi / idn[None, :]
    ref = np.concatenate([np.sin(arg), np.cos(arg), fixed.double().cpu().numpy()[:, None], np.sin(ia), np.cos(ia)], 1)
    # (float32 sin / cos of arguments up to 1e4: the argument's own rounding, 1e4 * 2^-24, bounds the agreement)
    assert float(np.abs(a.double().cpu().numpy() - ref).max()) < 2e-3
    assert float(np.abs(a.double().cpu().numpy()[:, 32:] - ref[:, 32:]).max()) < 2e-5


def _edge_embed_strided(dev, B, N, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    R, P = B * N, B * N * N
    W0, W2, W4 = rn(128, 120, sc=0.1), rn(128, 128, sc=0.1), rn(128, 128, sc=0.1)
    img = ops.edge_embed_pack(W0, W2, W4)
    seq = torch.arange(N).repeat(B, 1).reshape(-1).to(torch.int64).to(dev)
    sc_ca = rn(R, 3, sc=5.0)
    _, idenom, lower, upper = ops.feature_tables(torch.device(dev))
    pq = rn(R, 256, sc=0.5)
    p, q = pq[:, :128].contiguous(), pq[:, 128:].contiguous()
    b2, b3, gm, bt = rn(128, sc=0.2), rn(128, sc=0.2), 1 + rn(128, sc=0.1), rn(128, sc=0.1)
    outs = []
    for strided in (False, True):
        out = torch.full((P, 128), float("nan"), device=dev)
        if strided:
            ops.edge_embed(seq, sc_ca, idenom, lower, upper, img, pq, pq[:, 128:], b2, b3, gm, bt, out, P, N, ld_pq=256)
        else:
            ops.edge_embed(seq, sc_ca, idenom, lower, upper, img, p, q, b2, b3, gm, bt, out, P, N)
        outs.append(out)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_node_feats_ld_emu(use_emu):
    _node_feats("cpu", 2, 9, 0)


def test_edge_embed_ld_pq_emu(use_emu):
    _edge_embed_strided("cpu", 1, 9, 1)


@pytest.mark.gpu
def test_embed_feats_gpu(hip_lib):
    _node_feats("cuda", 3, 50, 2)
    _edge_embed_strided("cuda", 2, 40, 3)
    _edge_embed_strided("cuda", 1, 128, 4)
