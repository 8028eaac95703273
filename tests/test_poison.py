"""Every scratch / output tensor the hot path allocates with torch.empty is fully written before it is read: with the
allocator of ops / network / trunk poisoned (NaN fills instead of uninitialised memory) a training step -- forward, fused DSM
loss, backward -- still yields a finite loss and finite gradients, on the launch-sequence path and with the one-launch IPA
attention kernels forced.  (Guards changes like `dproj = empty(...)`: a column nobody assigns would surface here.)"""
import torch

from se3_diffusion_amd import loss as floss, network as nw, ops, options, train_step as ts, trunk
from se3_diffusion_amd.model.score_network import ScoreNetwork


def _poison(shape, like, dtype=torch.float32):
    if dtype.is_floating_point:
        return torch.full(tuple(shape), float("nan"), device=like.device, dtype=dtype)
    return torch.empty(shape, device=like.device, dtype=dtype)


def _run(dev, B, N, monkeypatch):
    for mod in (ops, nw, trunk):
        monkeypatch.setattr(mod, "empty", _poison)
    model = ScoreNetwork(ts.base_model_conf(2), diffuser=None).to(dev).train()      # (two blocks: one edge transition)
    ts.perturb_final_layers(model, seed=0)
    batch = ts.synthetic_batch(B, N, dev, seed=1)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])
    for kw in (dict(), dict(flash_ipa_min_tiles=0, flash_ipa_bwd_min_tiles=0)):
        with options.override(**kw):
            model.zero_grad()
            loss = floss.dsm_loss(batch, model(batch), gt37)
            loss.backward()
            assert bool(torch.isfinite(loss.detach())), kw
            bad = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            assert not bad, (kw, bad[:5])


def test_poisoned_allocations_emu(use_emu, monkeypatch):
    _run("cpu", 1, 12, monkeypatch)
