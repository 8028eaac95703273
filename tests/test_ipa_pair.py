"""Fused IPA pair pass (csrc/fd_ipa_pair.hip) against the unfused kernel sequence it replaces (fd_gemm z -> zb,
fd_ipa_softmax_fwd, fd_ipa_opair_fwd; fd_ipa_opair_bwd, fd_ipa_softmax_bwd, dz += dzb W40, dW40 = dzb^T z), which the
oracle parity tests pin to the reference (model/ipa_pytorch.py:380-422,455-457).  Both compute in fp32 with different
summation orders: 2e-5 of each tensor's maximum."""
import math

import pytest
import torch

from se3_diffusion_amd import ops
from se3_diffusion_amd.ops import lib, mv

H, PQ, ZB, CZ, LDF, F_PAIR = 8, 8, 40, 128, 2688, 2432


def _inputs(dev, B, N, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    R, P = B * N, B * N * N
    mask = torch.ones(B, N)
    mask[:, -2:] = 0                                    # padded residues
    return dict(z=rn(P, CZ), W40=rn(ZB, CZ, sc=0.1), b40=rn(ZB, sc=0.1), qp=rn(R, H, PQ * 3), kp=rn(R, H, PQ * 3),
                hw=rn(H, sc=0.5), mask=mask.reshape(-1).to(dev), S0=rn(B, H, N, N), dfeats=rn(R, LDF),
                dA0=rn(B, H, N, N), dz0=rn(P, CZ))


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def _points_soa(dev, B, N, seed):
    """fd_ipa_points_fwd's second copy of the key points ([B, 8, 24, N]) holds exactly kp's values."""
    g = torch.Generator().manual_seed(seed)
    R = B * N
    proj = torch.randn(R, 6816, generator=g).to(dev)
    quat = torch.nn.functional.normalize(torch.randn(R, 4, generator=g), dim=-1).to(dev)
    trans = torch.randn(R, 3, generator=g).to(dev)
    out = [torch.empty(R, H, n * 3, device=dev) for n in (PQ, PQ, 12)]
    ref = [torch.empty(R, H, n * 3, device=dev) for n in (PQ, PQ, 12)]
    kpT = torch.full((B, H, PQ * 3, N), float("nan"), device=dev)
    lib().call("fd_ipa_points_fwd", proj, quat, trans, *ref, None, 0, R, H, 256, PQ, 12)
    lib().call("fd_ipa_points_fwd", proj, quat, trans, *out, kpT, N, R, H, 256, PQ, 12)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    assert torch.equal(kpT, ref[1].reshape(B, N, H, PQ * 3).permute(0, 2, 3, 1))


def _run(dev, B, N, seed=0):
    t = _inputs(dev, B, N, seed)
    L = lib()
    R, P = B * N, B * N * N
    e = lambda *s: torch.empty(*s, device=dev)
    zer = lambda *s: torch.zeros(*s, device=dev)
    _points_soa(dev, B, N, seed)
    # ---- forward ----
    zb = e(P, ZB)
    ops.linear(mv(t["z"]), mv(t["W40"]), t["b40"], mv(zb), P, ZB, CZ)
    S_u = t["S0"].clone()
    L.call("fd_ipa_softmax_fwd", S_u, zb, t["qp"], t["kp"], t["hw"], t["mask"], B, N)
    f_u = zer(R, LDF)
    L.call("fd_ipa_opair_fwd", S_u, zb, f_u, B, N)
    # softmax + o_pair in one launch (fd_ipa_attn_fwd, the shipped path): the same arithmetic, bit for bit
    S_a = t["S0"].clone()
    f_a = zer(R, LDF)
    L.call("fd_ipa_attn_fwd", S_a, zb, t["qp"], t["kp"], None, t["hw"], t["mask"], f_a, B, N)
    assert torch.equal(S_a, S_u) and torch.equal(f_a, f_u)
    # ... and with the key points read from the [B, 8, 24, N] copy (what the network passes)
    kpT = t["kp"].reshape(B, N, H, PQ * 3).permute(0, 2, 3, 1).contiguous()
    S_a = t["S0"].clone()
    f_a = zer(R, LDF)
    L.call("fd_ipa_attn_fwd", S_a, zb, t["qp"], t["kp"], kpT, t["hw"], t["mask"], f_a, B, N)
    assert torch.equal(S_a, S_u) and torch.equal(f_a, f_u)
    S_f = t["S0"].clone()
    f_f = zer(R, LDF)
    L.call("fd_ipa_pair_fwd", S_f, t["z"], t["W40"], t["b40"], t["qp"], t["kp"], t["hw"], t["mask"], f_f, B, N)
    assert rel(S_f, S_u) < 2e-5
    assert rel(f_f[:, F_PAIR:F_PAIR + 256], f_u[:, F_PAIR:F_PAIR + 256]) < 2e-5
    assert float(f_f[:, :F_PAIR].abs().max()) == 0.0
    # ---- backward ----
    dA_u = t["dA0"].clone()
    dzb = e(P, ZB)
    L.call("fd_ipa_opair_bwd", S_u, zb, t["dfeats"], dA_u, dzb, B, N)
    dqp_u, dkp_u, dhw_u, part = e(R, H, PQ * 3), e(R, H, PQ * 3), zer(H), e(R, H)
    L.call("fd_ipa_softmax_bwd", S_u, dA_u, t["qp"], t["kp"], t["hw"], dzb, dqp_u, dkp_u, dhw_u, part, B, N)
    # o_pair backward + softmax backward in one launch (fd_ipa_attn_bwd): bit-identical to the pair of launches
    dA_a, dzb_a = t["dA0"].clone(), e(P, ZB)
    dqp_a, dkp_a, dhw_a, part_a = e(R, H, PQ * 3), e(R, H, PQ * 3), zer(H), e(R, H)
    for soa in (None, kpT):
        dA_a, dzb_a = t["dA0"].clone(), e(P, ZB)
        dqp_a, dkp_a, dhw_a, part_a = e(R, H, PQ * 3), e(R, H, PQ * 3), zer(H), e(R, H)
        L.call("fd_ipa_attn_bwd", S_u, dA_a, zb, t["dfeats"], t["qp"], t["kp"], soa, t["hw"], dzb_a, dqp_a, dkp_a, dhw_a,
               part_a, B, N)
        for name, got, want in (("dA", dA_a, dA_u), ("dzb", dzb_a, dzb), ("dqp", dqp_a, dqp_u), ("dkp", dkp_a, dkp_u)):
            assert torch.equal(got, want), (name, soa is not None)
        assert rel(dhw_a, dhw_u) < 1e-5          # (column sums of per-row partials: atomics, order not fixed)
    dz_u = t["dz0"].clone()
    ops.linear_dx(mv(dzb), mv(t["W40"]), mv(dz_u), P, ZB, CZ, beta=True)
    dW_u, db_u = zer(ZB, CZ), zer(ZB)
    ops.linear_dw(mv(dzb), mv(t["z"]), mv(dW_u), P, ZB, CZ)
    ops.bias_grad(mv(dzb), db_u, P, ZB)
    for acc in (1, 0):
        dA_f = t["dA0"].clone()
        dz_f = t["dz0"].clone()
        dqp_f, dkp_f, dhw_f, part_f, dW_f, db_f = e(R, H, PQ * 3), e(R, H, PQ * 3), zer(H), e(R, H), zer(ZB, CZ), zer(ZB)
        L.call("fd_ipa_pair_bwd", S_u, dA_f, t["z"], t["W40"], t["b40"], t["dfeats"], t["qp"], t["kp"], t["hw"], dz_f, acc,
               dqp_f, dkp_f, dhw_f, part_f, dW_f, db_f, B, N)
        assert rel(dA_f, dA_u) < 2e-5
        assert rel(dz_f, dz_u if acc else dz_u - t["dz0"]) < 2e-5
        assert rel(dqp_f, dqp_u) < 2e-5 and rel(dkp_f, dkp_u) < 2e-5 and rel(dhw_f, dhw_u) < 2e-5
        assert rel(dW_f, dW_u) < 2e-5 and rel(db_f, db_u) < 2e-5


def test_ipa_pair_emu(use_emu):
    _run("cpu", B=1, N=70)        # two staged chunks, the second ragged; padded residues


@pytest.mark.gpu
def test_ipa_pair_gpu(hip_lib):
    _run("cuda", B=2, N=70)
    _run("cuda", B=3, N=128, seed=1)
    _run("cuda", B=1, N=200, seed=2)       # NMAX = 256 instantiation
    _run("cuda", B=1, N=512, seed=3)       # NMAX = 512
    _run("cuda", B=5, N=150, seed=4)       # more rows than persistent blocks -> several rows per block


def _dz_acc(dev, rows, seed=0):
    """dz (+)= dzb W40 (fd_ipa_dz_acc) against float64, accumulate and assign, ragged last tile"""
    g = torch.Generator().manual_seed(seed)
    dzb, W40, dz0 = (torch.randn(rows, ZB, generator=g).to(dev), (torch.randn(ZB, CZ, generator=g) * 0.1).to(dev),
                     torch.randn(rows, CZ, generator=g).to(dev))
    ref = dzb.double().cpu() @ W40.double().cpu()
    for acc in (1, 0):
        dz = dz0.clone()
        lib().call("fd_ipa_dz_acc", dzb, W40, dz, rows, acc)
        want = ref + (dz0.double().cpu() if acc else 0)
        assert float((dz.double().cpu() - want).abs().max() / want.abs().max()) < 2e-6


def test_ipa_dz_acc_emu(use_emu):
    _dz_acc("cpu", 100)
    _dz_acc("cpu", 32 * 9, seed=1)


@pytest.mark.gpu
def test_ipa_dz_acc_gpu(hip_lib):
    _dz_acc("cuda", 100)
    _dz_acc("cuda", 70 * 70 * 2, seed=1)
    _dz_acc("cuda", 30 * 128 * 128, seed=2)


def _zb(dev, rows, seed=0):
    """zb = z W40^T + b40 (fd_ipa_zb) against float64, ragged last tile"""
    g = torch.Generator().manual_seed(seed)
    z, W40, b40 = torch.randn(rows, CZ, generator=g).to(dev), (torch.randn(ZB, CZ, generator=g) * 0.1).to(dev), torch.randn(ZB, generator=g).to(dev)
    zb = torch.full((rows, ZB), 7.0, device=dev)
    lib().call("fd_ipa_zb", z, W40, b40, zb, rows)
    ref = z.double().cpu() @ W40.double().cpu().T + b40.double().cpu()
    assert float((zb.double().cpu() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_ipa_zb_emu(use_emu):
    _zb("cpu", 100)


@pytest.mark.gpu
def test_ipa_zb_gpu(hip_lib):
    _zb("cuda", 100)
    _zb("cuda", 70 * 70 * 2, seed=1)
    _zb("cuda", 30 * 128 * 128, seed=2)
