"""fd_dsm_loss (fused Experiment.loss_fn arithmetic, value + gradient) against the plain-torch formulation of the same
loss (train_step.dsm_loss, a line-by-line restatement of experiments/train_se3_diffusion.py:538-666), in float64.
CPU tier = SIMT interpreter; GPU tier = gfx950."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from se3_diffusion_amd import loss as floss
from se3_diffusion_amd import train_step as ts

# outputs + gradients of the UNMODIFIED reference's Experiment.loss_fn, both rotation branches (oracle/make_golden_loss.py)
GL = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss.npz"))


def _case(B, N, seed, dev, n_pad=0, n_fixed=0, t_values=None):
    g = torch.Generator().manual_seed(seed)
    batch = ts.synthetic_batch(B, N, "cpu", seed=seed)
    if t_values is not None:
        batch["t"] = torch.tensor(t_values, dtype=torch.float32)
    if n_pad:
        batch["res_mask"][:, N - n_pad:] = 0
    if n_fixed:
        batch["fixed_mask"][:, :n_fixed] = 1
    gt37 = torch.randn(B, N, 37, 3, generator=g) * 3
    gt37[0, 1, 2] = 0          # an all-zero ground-truth atom drops out of the atom mask
    out = dict(rot_score=torch.randn(B, N, 3, generator=g, dtype=torch.float64),
               trans_score=torch.randn(B, N, 3, generator=g),
               rigids=torch.cat([torch.randn(B, N, 4, generator=g), batch["rigids_0"][..., 4:] + torch.randn(B, N, 3, generator=g)], -1),
               atom37=gt37 + 0.7 * torch.randn(B, N, 37, 3, generator=g))
    out["rot_score"][0, 0] = 0   # |pred| = 0: zero sub-gradient of the norm
    return batch, out, gt37


def _check(dev, B, N, seed, separate=True, **kw):
    batch, out, gt37 = _case(B, N, seed, dev, **kw)
    exp = SimpleNamespace(**dict(vars(ts.EXP), separate_rot_loss=separate))
    # reference: the torch formulation in float64
    ref_out = {k: v.double().clone().requires_grad_(True) for k, v in out.items()}
    ref_batch = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    ref = ts.dsm_loss(ref_batch, ref_out, gt37.double(), exp)
    ref.backward()
    # fused kernels
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    dout = {k: v.to(dev).clone().requires_grad_(True) for k, v in out.items()}
    loss, terms = floss.dsm_loss(dbatch, dout, gt37.to(dev), exp, with_terms=True)
    loss.backward()
    lv, rv = float(loss.detach()), float(ref.detach())
    assert abs(lv - rv) < 2e-5 * abs(rv) + 1e-6, (lv, rv)
    for k in ("rot_score", "trans_score", "rigids", "atom37"):
        want = ref_out[k].grad
        got = dout[k].grad.cpu().double()
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) < 2e-4 * scale + 1e-9, (k, float((got - want).abs().max()), scale)
    assert torch.isfinite(terms["final"]).all()


def _golden_inputs(dev):
    batch = {k[6:]: torch.tensor(GL[k]).to(dev) for k in GL.files if k.startswith("batch/")}
    outs = {k[4:]: torch.tensor(GL[k]).to(dev).requires_grad_(True) for k in GL.files if k.startswith("out/")}
    return batch, outs, torch.tensor(GL["gt_atom37"]).to(dev)


def _exp(tag):
    return SimpleNamespace(**dict(vars(ts.EXP), separate_rot_loss=(tag == "sep")))


@pytest.mark.parametrize("tag", ["sep", "joint"])
def test_torch_restatement_matches_reference_loss_fn(tag):
    """train_step.dsm_loss (the checker of the fused kernel) == the reference's loss_fn, value and gradients"""
    batch, outs, gt37 = _golden_inputs("cpu")
    loss = ts.dsm_loss(batch, outs, gt37, _exp(tag))
    loss.backward()
    assert abs(float(loss.detach()) - float(GL[f"{tag}/loss"])) < 1e-6 * abs(float(GL[f"{tag}/loss"]))
    for k, v in outs.items():
        want = GL[f"{tag}/grad/{k}"]
        assert np.abs(v.grad.numpy() - want).max() < 1e-5 * np.abs(want).max() + 1e-9, k


def _golden_fused(dev, tag):
    batch, outs, gt37 = _golden_inputs(dev)
    loss, terms = floss.dsm_loss(batch, outs, gt37, _exp(tag), with_terms=True)
    loss.backward()
    want = float(GL[f"{tag}/loss"])
    assert abs(float(loss.detach()) - want) < 2e-5 * abs(want)
    assert np.allclose(terms["final"].cpu().numpy(), GL[f"{tag}/batch_train_loss"], rtol=2e-5, atol=1e-6)
    assert np.allclose((terms["axis_loss"] + terms["angle_loss"]).cpu().numpy(), GL[f"{tag}/batch_rot_loss"], rtol=2e-5, atol=1e-6)
    assert np.allclose(terms["bb_atom_loss"].cpu().numpy(), GL[f"{tag}/batch_bb_atom_loss"], rtol=2e-5, atol=1e-6)
    assert np.allclose(terms["dist_mat_loss"].cpu().numpy(), GL[f"{tag}/batch_dist_mat_loss"], rtol=2e-5, atol=1e-6)
    for k, v in outs.items():
        g = GL[f"{tag}/grad/{k}"]
        assert np.abs(v.grad.cpu().numpy() - g).max() < 2e-4 * np.abs(g).max() + 1e-9, k


@pytest.mark.parametrize("tag", ["sep", "joint"])
def test_dsm_loss_golden_emu(use_emu, tag):
    _golden_fused("cpu", tag)


def test_dsm_loss_emu(use_emu):
    _check("cpu", 3, 9, 1, t_values=[0.1, 0.22, 0.9])            # both sides of the 0.2 / 0.25 filters
    _check("cpu", 2, 12, 2, n_pad=3, n_fixed=2, t_values=[0.05, 0.6])
    _check("cpu", 2, 12, 2, n_pad=3, n_fixed=2, t_values=[0.05, 0.6], separate=False)


def test_dsm_loss_matches_float32_training_loss_emu(use_emu):
    """drop-in for train_step.dsm_loss on the actual fp32 tensors of a training step"""
    batch, out, gt37 = _case(2, 8, 5, "cpu", t_values=[0.15, 0.7])
    a = float(ts.dsm_loss(batch, out, gt37))
    b = float(floss.dsm_loss(batch, out, gt37))
    assert abs(a - b) < 1e-4 * abs(a)


@pytest.mark.gpu
def test_dsm_loss_gpu(hip_lib):
    _golden_fused("cuda", "sep")
    _golden_fused("cuda", "joint")
    _check("cuda", 3, 40, 6, n_pad=4, n_fixed=3, t_values=[0.05, 0.24, 0.6], separate=False)
    _check("cuda", 4, 128, 3, t_values=[0.1, 0.22, 0.9, 0.5])
    _check("cuda", 3, 70, 4, n_pad=9, n_fixed=5, t_values=[0.05, 0.24, 0.6])
