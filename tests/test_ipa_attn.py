"""The per-row IPA attention forward (fd_ipa_attn_fwd: logits + softmax + o_pair of a query row, model/ipa_pytorch.py:380-457
without the q k^T / a v products) in its two forms -- zb row block through an LDS image (N <= 256), direct reads (longer rows) --
against the separate softmax and o_pair kernels, and against float64."""
import math

import pytest
import torch

from se3_diffusion_amd import ops

H, PQ, ZB, CZ4, LDF = 8, 8, 40, 32, 2688


def _run(dev, B, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    R = B * N
    S0 = rn(B, H, N, N)
    zb = rn(R * N, ZB)
    qp, kp = rn(R, H, PQ * 3), rn(R, H, PQ * 3)
    kpT = kp.view(B, N, H, PQ * 3).permute(0, 2, 3, 1).contiguous()
    hw = rn(H)
    mask = (torch.rand(R, generator=g) > 0.1).float().to(dev)
    L = ops.lib()
    # fused
    S1 = S0.clone()
    f1 = torch.zeros(R, LDF, device=dev)
    L.call("fd_ipa_attn_fwd", S1, zb, qp, kp, kpT, hw, mask, f1, B, N)
    # separate kernels
    S2 = S0.clone()
    f2 = torch.zeros(R, LDF, device=dev)
    L.call("fd_ipa_softmax_fwd", S2, zb, qp, kp, hw, mask, B, N)
    L.call("fd_ipa_opair_fwd", S2, zb, f2, B, N)
    assert float((S1 - S2).abs().max()) < 2e-6
    nz = f2.abs().sum(0) > 0
    assert int(nz.sum()) == H * CZ4            # (the o_pair columns, wherever they sit)
    assert float((f1 - f2).abs().max()) < 2e-5 * float(f2.abs().max())
    # float64 restatement of the logits / softmax
    gsc = math.sqrt(1.0 / (3.0 * (PQ * 9.0 / 2.0)))
    gamma = torch.nn.functional.softplus(hw.double().cpu()) * gsc
    q = qp.double().cpu().view(B, N, H, PQ, 3)
    k = kp.double().cpu().view(B, N, H, PQ, 3)
    d2 = ((q[:, :, None] - k[:, None]) ** 2).sum((-1, -2))                    # [B, i, j, H]
    m = mask.double().cpu().view(B, N)
    logit = (S0.double().cpu() + math.sqrt(1 / 3) * zb.double().cpu().view(B, N, N, ZB)[..., :H].permute(0, 3, 1, 2)
             - 0.5 * (d2 * gamma).permute(0, 3, 1, 2) + 1e5 * (m[:, None, :, None] * m[:, None, None, :] - 1))
    ref = torch.softmax(logit, -1)
    # (rows of masked residues: every logit carries -1e5 in fp32, as in the reference -- their probabilities are fp32 round-off
    # of that shift and are multiplied by the residue mask downstream; compared on the unmasked rows)
    rows = (m > 0)[:, None, :, None]
    assert float(((S1.double().cpu() - ref) * rows).abs().max()) < 2e-5


def test_ipa_attn_fwd_emu(use_emu):
    _run("cpu", 1, 12)
    _run("cpu", 1, 260, seed=1)        # direct reads


@pytest.mark.gpu
def test_ipa_attn_fwd_gpu(hip_lib):
    _run("cuda", 2, 128)
    _run("cuda", 1, 256, seed=1)
    _run("cuda", 1, 257, seed=2)
    _run("cuda", 2, 400, seed=3)
    _run("cuda", 1, 512, seed=4)
    _run("cuda", 1, 600, seed=5)
