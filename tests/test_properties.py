"""Size-independent properties of the hot path at BASELINE.json's FULL sizes (B=30 x N=128, N=256, N=512), where the
oracle is too slow to be the checker (gfx950 only, -m gpu):

  * SE(3) equivariance of ScoreNetwork.forward: a global rotation (+ translation) of the input frames rotates the predicted
    frames / atoms / translation score and leaves the rotation score and psi unchanged (IPA is invariant by construction,
    ipa_pytorch.py:303-471; the translation score is only rotation-equivariant because the R^3 diffuser assumes centred data);
  * batch independence: an example's outputs do not depend on its batch mates;
  * GEMM checksums at the pair-level size (M = 491,520): column sums of C against (column sums of A) W^T in float64, and
    linearity in W, through the persistent split-bf16 kernel and the fp32-MFMA kernel;
  * backward at full size: the directional derivative of the training loss along a random parameter direction, by
    central differences of the forward pass, against <grad, direction> from the hand-written backward;
  * diffuser round trips: x_0 recovered from (trans_score, x_t, t) (r3_diffuser.py:45-50) and the rotation score of
    forward_marginal aligned with the rotation it applied.

Tolerances are fp32 round-off of a 4-block network on coordinates of +-100 A: 2e-3 of each tensor's max magnitude.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import framediff_oracle as fo  # noqa: E402  (input synthesis + quaternion algebra on the host)
from se3_diffusion_amd import trunk  # noqa: E402
from se3_diffusion_amd.data import se3_diffuser  # noqa: E402

pytestmark = pytest.mark.gpu
CACHE = os.environ.get("FD_TEST_IGSO3_CACHE", "/tmp/fd_test_igso3_cache")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _forward(P, feats, blocks=4):
    with torch.no_grad():
        out, _ = trunk.forward(P, {k: v.cuda() for k, v in feats.items()}, blocks, save=False)
    return {k: v.cpu() for k, v in out.items()}


def _params(blocks=4, seed=0):
    return {k: v.cuda() for k, v in fo.synth_params(seed=seed, conf=dict(fo.CONF, num_blocks=blocks)).items()}


def _moved(feats, qg, tg):
    """the same backbone after the global rigid motion x -> R(qg) x + tg"""
    f = {k: v.clone() for k, v in feats.items()}
    Rg = fo.quat_to_rot(qg)
    q = fo.quat_multiply(qg.expand_as(f["rigids_t"][..., :4]), f["rigids_t"][..., :4])
    t = fo.rot_apply(Rg, f["rigids_t"][..., 4:]) + tg
    f["rigids_t"] = torch.cat([q, t], -1)
    f["sc_ca_t"] = fo.rot_apply(Rg, f["sc_ca_t"]) + tg
    return f


@pytest.mark.parametrize("B,N", [(30, 128), (7, 256), (2, 512)])
def test_se3_equivariance_full_size(hip_lib, B, N):
    P = _params()
    feats = fo.synth_feats(B, N, seed=11, n_pad=5, n_fixed=3)
    g = torch.Generator().manual_seed(5)
    qg = torch.randn(4, generator=g)
    qg = qg / qg.norm()
    Rg = fo.quat_to_rot(qg)
    base = _forward(P, feats)
    # rotation only: everything transforms
    rot = _forward(P, _moved(feats, qg, torch.zeros(3)))
    m = feats["res_mask"][..., None]
    assert rel(rot["rot_score"], base["rot_score"]) < 2e-3
    assert rel(rot["psi"], base["psi"]) < 2e-3
    assert rel(rot["trans_score"], fo.rot_apply(Rg, base["trans_score"])) < 2e-3
    assert rel(rot["atom37"][:, :, :5] * m[..., None], fo.rot_apply(Rg, base["atom37"][:, :, :5]) * m[..., None]) < 2e-3
    want_q = fo.quat_multiply(qg.expand_as(base["rigids"][..., :4]), base["rigids"][..., :4])
    sgn = torch.sign((want_q * rot["rigids"][..., :4]).sum(-1, keepdim=True))
    assert rel(rot["rigids"][..., :4] * sgn * m, want_q * m) < 2e-3
    # rotation + translation: frames and atoms follow, the rotation score and psi do not move
    tg = torch.tensor([7.0, -4.0, 11.0])
    mv = _forward(P, _moved(feats, qg, tg))
    assert rel(mv["rot_score"], base["rot_score"]) < 2e-3
    assert rel(mv["psi"], base["psi"]) < 2e-3
    assert rel(mv["rigids"][..., 4:] * m, (fo.rot_apply(Rg, base["rigids"][..., 4:]) + tg) * m) < 2e-3
    assert rel(mv["atom37"][:, :, :5] * m[..., None], (fo.rot_apply(Rg, base["atom37"][:, :, :5]) + tg) * m[..., None]) < 2e-3


def test_batch_independence_full_size(hip_lib):
    P = _params()
    B, N = 30, 128
    feats = fo.synth_feats(B, N, seed=12)
    full = _forward(P, feats)
    for b in (0, 17, 29):
        one = _forward(P, {k: v[b:b + 1] for k, v in feats.items()})
        for k in ("rot_score", "trans_score", "psi", "atom37"):
            assert rel(one[k], full[k][b:b + 1]) < 2e-3, (b, k)
    # NB padding invariance is NOT a property of the reference: ipa_pytorch.py:636-637 hands nn.TransformerEncoder the FLOAT
    # tensor 1 - mask as src_key_padding_mask, which the training path adds to the logits (+1 on padded keys) instead of
    # masking them, so padded residues take part in the sequence attention (the oracle restates exactly that:
    # tfmr_mask_mode="additive"; with 4 padded residues its outputs move by 7-40 %).


def test_directional_derivative_full_size(hip_lib):
    """fwd + fused DSM loss + bwd of the B=30 x N=128 training step (the bench.py workload)"""
    from se3_diffusion_amd import loss as floss
    from se3_diffusion_amd import train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    B, N = 30, 128
    m = ScoreNetwork(ts.base_model_conf(4), diffuser=None)
    m.load_state_dict(fo.synth_params(seed=2, conf=dict(fo.CONF, num_blocks=4)), strict=True)
    m = m.cuda().train()
    batch = ts.synthetic_batch(B, N, "cuda", seed=7)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])
    loss = floss.dsm_loss(batch, m(batch), gt37)
    loss.backward()
    g = torch.Generator(device="cuda").manual_seed(4)
    params = [p for p in m.parameters() if p.grad is not None]
    dirs = [torch.randn(p.shape, device="cuda", generator=g) * p.detach().abs().mean() for p in params]
    analytic = sum(float((p.grad.double() * d.double()).sum()) for p, d in zip(params, dirs))

    def loss_at(eps):
        with torch.no_grad():
            for p, d in zip(params, dirs):
                p.add_(d, alpha=eps)
            v = float(floss.dsm_loss(batch, m(batch), gt37))
            for p, d in zip(params, dirs):
                p.sub_(d, alpha=eps)
        return v

    # the step is chosen so that the loss moves by ~1e-3 of its value: far above fp32 forward noise, small enough for
    # the second-order term of a central difference
    lv = float(loss.detach())
    eps = 1e-3 * abs(lv) / (abs(analytic) + 1e-12)
    eps = min(max(eps, 1e-4), 5e-2)
    fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
    assert abs(fd - analytic) < 3e-2 * abs(analytic) + 1e-6, (fd, analytic, eps, lv)


@pytest.mark.parametrize("tile", [0, 1])
def test_gemm_checksums_pair_level_size(hip_lib, tile):
    """M = B N^2 = 491,520 rows, N = K = 384 (the edge-transition GEMM of the B=30 x N=128 step)"""
    M, N, K = 30 * 128 * 128, 384, 384
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randn(M, K, device="cuda", generator=g)
    W1 = torch.randn(N, K, device="cuda", generator=g) * 0.05
    W2 = torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)

    def run(W, b=None):
        C = torch.empty(M, N, device="cuda")
        hip_lib.gemm(A, W, C, M, N, K, (K, 1), (1, K), N, bias=b, tile=tile)
        return C

    C1, C2, C12 = run(W1), run(W2), run(W1 + W2)
    # checksum of checksums: column sums of C == (column sums of A) W^T, in float64
    want = A.double().sum(0) @ W1.double().t()
    got = C1.double().sum(0)
    assert float((got - want).abs().max()) < 1e-6 * float(C1.double().abs().sum(0).max())
    # linearity in W (fp32 round-off of W1 + W2 and of the accumulation only)
    assert float((C12 - (C1 + C2)).abs().max()) < 2e-5 * float(C12.abs().max())
    # a row block at the far end against float64
    ref = A[-512:].double() @ W1.double().t() + bias.double()
    assert float((run(W1, bias)[-512:].double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())


def test_diffuser_round_trips_full_size(hip_lib):
    ns = SimpleNamespace
    diff = se3_diffuser.SE3Diffuser(ns(diffuse_trans=True, diffuse_rot=True, r3=ns(min_b=0.1, max_b=20.0, coordinate_scaling=0.1),
                                       so3=ns(num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5, schedule="logarithmic",
                                              cache_dir=CACHE, use_cached_score=False)))
    B, N = 8, 512
    rig0 = fo.synth_feats(B, N, seed=13)["rigids_t"].cuda()
    t = np.linspace(0.05, 0.95, B)
    gen = torch.Generator(device="cuda").manual_seed(1)
    out = diff.forward_marginal_batch(rig0, t, generator=gen)
    r3 = diff._r3_diffuser
    # x_0 = (score * var + x_t) / exp(-b/2) in scaled units (calc_trans_0)
    beta = torch.tensor(r3.marginal_b_t(t), dtype=torch.float64, device="cuda")[:, None, None]
    xt = out["rigids_t"][..., 4:].double() * 0.1
    x0 = (out["trans_score"].double() * (1 - torch.exp(-beta)) + xt) / torch.exp(-0.5 * beta)
    assert float((x0 - rig0[..., 4:].double() * 0.1).abs().max()) < 2e-5 * float(xt.abs().max())
    # the rotation that was applied, R_0^T R_t, has the axis of the returned score (and the score points back: g < 0)
    q0 = rig0[..., :4].cpu()
    q0 = q0 / q0.norm(dim=-1, keepdim=True)
    rel_q = fo.quat_multiply(fo.invert_quat(q0), out["rigids_t"][..., :4].cpu())
    v = fo.quat_to_rotvec(rel_q)
    s = out["rot_score"].cpu()
    cosang = (v * s).sum(-1) / (v.norm(dim=-1) * s.norm(dim=-1) + 1e-12)
    big = (v.norm(dim=-1) > 0.05) & (s.norm(dim=-1) > 1e-3)
    assert big.float().mean() > 0.9 and float(cosang[big].abs().min()) > 0.999
