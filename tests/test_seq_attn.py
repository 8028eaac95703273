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
