"""Fused sequence-transformer attention (csrc/fd_seq_attn.hip) against float64: softmax(q k^T / sqrt(d) + key mask) v per
(batch, head) of torch.nn.TransformerEncoderLayer.self_attn as IpaScore configures it (model/ipa_pytorch.py:584-593:
nhead 4, d_model 320).  fp32 MFMA contractions: 3e-6 of the tensor maximum."""
import math

import pytest
import torch

from se3_diffusion_amd.ops import lib

TH, THD, TD = 4, 80, 320


def _run(dev, B, N, seed=0, inf_mask=False, want_A=True):
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(B * N, 3 * TD, generator=g)
    mask = torch.ones(B, N)
    mask[:, N - 3:] = 0                                   # padded residues at the end of every backbone
    # training-mode quirk of the reference (additive float mask 1 - mask) or the eval-mode boolean semantics (-inf)
    key_add = torch.where(mask > 0, torch.zeros_like(mask), torch.full_like(mask, float("-inf"))) if inf_mask else 1 - mask
    out = torch.empty(B * N, TD, device=dev)
    A = torch.empty(B, TH, N, N, device=dev) if want_A else None
    lib().call("fd_seq_attn_fwd", qkv.to(dev), key_add.to(dev), out, A, 1.0 / math.sqrt(THD), B, N)
    q, k, v = (qkv.double().view(B, N, 3, TH, THD)[:, :, i].permute(0, 2, 1, 3) for i in range(3))   # [B,TH,N,THD]
    S = q @ k.transpose(-1, -2) / math.sqrt(THD) + key_add.double()[:, None, None, :]
    Ar = torch.softmax(S, -1)
    ref = (Ar @ v).permute(0, 2, 1, 3).reshape(B * N, TD)
    assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < 3e-6
    if want_A:
        assert float((A.cpu().double() - Ar).abs().max()) < 3e-6


def test_seq_attn_emu(use_emu):
    _run("cpu", B=2, N=37)                  # ragged query tile, ragged key tile, j padding to a multiple of 8
    _run("cpu", B=1, N=64, seed=1, inf_mask=True, want_A=False)


@pytest.mark.gpu
def test_seq_attn_gpu(hip_lib):
    _run("cuda", B=2, N=37)
    _run("cuda", B=3, N=128, seed=1)
    _run("cuda", B=2, N=128, seed=2, inf_mask=True, want_A=False)
    _run("cuda", B=1, N=300, seed=3)        # NMAX = 1024 instantiation
    _run("cuda", B=1, N=512, seed=4, want_A=False)


def _run_bwd(dev, B, N, seed=0):
    """fd_seq_attn_bwd (one launch: dQ, dK, dV from the saved probabilities and output) against float64 autograd of the same
    attention; untouched memory is checked through a poisoned output buffer."""
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(B * N, 3 * TD, generator=g)
    dout = torch.randn(B * N, TD, generator=g)
    mask = torch.ones(B, N)
    mask[:, N - 3:] = 0
    key_add = 1 - mask
    out = torch.empty(B * N, TD, device=dev)
    A = torch.empty(B, TH, N, N, device=dev)
    sc = 1.0 / math.sqrt(THD)
    lib().call("fd_seq_attn_fwd", qkv.to(dev), key_add.to(dev), out, A, sc, B, N)
    dqkv = torch.full((B * N, 3 * TD), float("nan"), device=dev)
    lib().call("fd_seq_attn_bwd", qkv.to(dev), A, dout.to(dev), out, dqkv, sc, B, N)
    x = qkv.double().clone().requires_grad_(True)
    q, k, v = (x.view(B, N, 3, TH, THD)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    S = q @ k.transpose(-1, -2) * sc + key_add.double()[:, None, None, :]
    o = (torch.softmax(S, -1) @ v).permute(0, 2, 1, 3).reshape(B * N, TD)
    (o * dout.double()).sum().backward()
    got = dqkv.cpu().double()
    assert bool(torch.isfinite(got).all())
    errs = {}
    for name, sl in (("dQ", slice(0, TD)), ("dK", slice(TD, 2 * TD)), ("dV", slice(2 * TD, 3 * TD))):
        ref = x.grad[:, sl]
        errs[name] = float((got[:, sl] - ref).abs().max() / ref.abs().max())
    assert max(errs.values()) < 5e-6, errs
    return errs


def test_seq_attn_bwd_emu(use_emu):
    _run_bwd("cpu", B=1, N=16)
    _run_bwd("cpu", B=2, N=37, seed=1)          # ragged tiles, N % 4 != 0, idle waves
    _run_bwd("cpu", B=1, N=80, seed=2)          # five tiles: a second group with three idle waves


@pytest.mark.gpu
def test_seq_attn_bwd_gpu(hip_lib):
    import parity_log
    with parity_log.case("seq_attn_bwd"):
        for (B, N, seed) in ((30, 128, 0), (3, 37, 1), (12, 200, 2), (7, 256, 3), (1, 301, 4), (2, 512, 5)):
            for k, e in _run_bwd("cuda", B, N, seed).items():
                parity_log.out(f"seq_attn_bwd.{k}", e)
