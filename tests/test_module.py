"""Drop-in boundary: se3_diffusion_amd.model.score_network.ScoreNetwork has the reference's
state_dict (names + shapes), runs through torch.autograd, and one training step (forward + DSM
loss + backward) agrees with the oracle.  CPU tier = SIMT interpreter; GPU tier = gfx950."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import framediff_oracle as fo  # noqa: E402
from se3_diffusion_amd import train_step as ts  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402


def test_state_dict_matches_reference_layout():
    m = ScoreNetwork(ts.base_model_conf(4), diffuser=None)
    sd = m.state_dict()
    want = fo.param_shapes(fo.CONF)            # pinned against the reference by make_golden (strict load)
    assert set(sd) == set(want)
    for k, shp in want.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sum(v.numel() for v in sd.values()) == 17446190


def _train_step(dev, B, N, blocks):
    conf = dict(fo.CONF, num_blocks=blocks)
    P = fo.synth_params(seed=11, conf=conf)
    m = ScoreNetwork(ts.base_model_conf(blocks), diffuser=None)
    m.load_state_dict(P, strict=True)
    m = m.to(dev).train()
    batch = ts.synthetic_batch(B, N, dev, seed=5)
    batch["t"][0] = 0.1   # exercise the t < 0.25 auxiliary losses
    cpu_batch = {k: v.cpu() for k, v in batch.items()}
    gt37, _ = fo.backbone_atoms(cpu_batch["rigids_0"][..., :4], cpu_batch["rigids_0"][..., 4:],
                                cpu_batch["torsion_angles_sin_cos"][..., 2, :])
    out = m(batch)
    loss = ts.dsm_loss(batch, out, gt37.to(dev))
    loss.backward()
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    oo = fo.score_network_forward(Po, cpu_batch, conf)
    lo = ts.dsm_loss(cpu_batch, oo, gt37)
    lo.backward()
    assert abs(float(loss) - float(lo)) < 1e-4 * abs(float(lo)) + 1e-6
    bad = []
    for n, p in m.named_parameters():
        gr = Po[n].grad if Po[n].grad is not None else torch.zeros_like(Po[n])
        g = p.grad.cpu() if p.grad is not None else torch.zeros_like(gr)
        err = float((g.double() - gr.double()).abs().max())
        if err > 3e-3 * float(gr.abs().max()) + 2e-5:
            bad.append((n, err, float(gr.abs().max())))
    assert not bad, bad[:8]


def test_train_step_emu(use_emu):
    _train_step("cpu", B=2, N=8, blocks=1)


def test_inplace_grad_accumulation_emu(use_emu):
    """accumulate_into_grad=True (bench.py / dist.FlatGrads) gives the same gradients as the autograd path."""
    from se3_diffusion_amd import dist as fdist
    conf = dict(fo.CONF, num_blocks=1)
    P = fo.synth_params(seed=12, conf=conf)
    batch = ts.synthetic_batch(1, 8, "cpu", seed=6)
    gt37, _ = fo.backbone_atoms(batch["rigids_0"][..., :4], batch["rigids_0"][..., 4:], batch["torsion_angles_sin_cos"][..., 2, :])
    grads = []
    for inplace in (False, True):
        m = ScoreNetwork(ts.base_model_conf(1), diffuser=None)
        m.load_state_dict(P, strict=True)
        m.train()
        if inplace:
            flat = fdist.FlatGrads(m.parameters())
            m.accumulate_into_grad = True
        ts.dsm_loss(batch, m(batch), gt37).backward()
        grads.append({n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()})
        if inplace:
            assert float(flat.flat.abs().sum()) > 0
    for n in grads[0]:
        assert torch.allclose(grads[0][n], grads[1][n], rtol=1e-4, atol=1e-6), n


def test_backbone_atoms_emu(use_emu):
    f = fo.synth_feats(2, 7, seed=3)
    psi = f["torsion_angles_sin_cos"][..., 2, :]
    a37, a14 = ts.backbone_atoms(f["rigids_t"], psi)
    r37, r14 = fo.backbone_atoms(f["rigids_t"][..., :4], f["rigids_t"][..., 4:], psi)
    assert (a37 - r37).abs().max() < 1e-4 and (a14 - r14).abs().max() < 1e-4


def test_eval_no_grad_path_emu(use_emu):
    conf = dict(fo.CONF, num_blocks=1)
    P = fo.synth_params(seed=2, conf=conf)
    m = ScoreNetwork(ts.base_model_conf(1), diffuser=None)
    m.load_state_dict(P)
    m.eval()
    feats = fo.synth_feats(1, 9, seed=2, n_pad=2)
    with torch.no_grad():
        out = m(feats)
        ref = fo.score_network_forward(P, feats, conf, tfmr_mask_mode="bool")
    for k in ("psi", "trans_score", "atom37"):
        assert (out[k] - ref[k]).abs().max() < 2e-4 * ref[k].abs().max() + 1e-6, k


@pytest.mark.gpu
def test_train_step_gpu(hip_lib):
    _train_step("cuda", B=3, N=24, blocks=2)
