"""Device-resident reverse diffusion: the loop of Experiment.inference_fn
(reference experiments/train_se3_diffusion.py:718-818) with the frames kept in HBM for all
num_t steps -- no device->numpy->scipy->eigh->device round trip per step
(reference :770-781, se3_diffuser.py:11-29) and B >= 1 equal-length backbones per launch.

Per step: ScoreNetwork forward (HIP trunk) -> self-conditioning CA update -> fd_se3_reverse_step.
The last step (t == min_t) takes the model's predicted frames directly (:778-780).
"""
from __future__ import annotations

import numpy as np
import torch

from .openfold.utils import rigid_utils as ru


def init_feats(diffuser, B, N, device, generator=None, noise=None):
    """Sampler.sample's data_init (inference_se3_diffusion.py:427-449) for B backbones of length N."""
    rig = diffuser.sample_ref_device(B * N, device, noise=noise, generator=generator).view(B, N, 7)
    dev = torch.device(device)
    return dict(
        res_mask=torch.ones(B, N, device=dev), fixed_mask=torch.zeros(B, N, device=dev),
        seq_idx=torch.arange(1, N + 1, device=dev)[None].repeat(B, 1),
        torsion_angles_sin_cos=torch.zeros(B, N, 7, 2, device=dev), sc_ca_t=torch.zeros(B, N, 3, device=dev),
        rigids_t=rig, t=torch.ones(B, device=dev))


@torch.no_grad()
def sample(model, diffuser, feats, num_t=500, min_t=0.01, noise_scale=0.1, self_condition=True, center=True,
           generator=None, noise_fn=None, return_traj=False):
    """Run the reverse process on `feats` (from init_feats).  noise_fn(step, shape) -> (z_rot, z_trans) injects
    draws (e.g. the numpy stream, for trajectory parity); default draws on the device.
    Returns dict(rigids [B,N,7], atom37 [B,N,37,3], psi, (rigid_traj list))."""
    feats = dict(feats)
    B, N = feats["res_mask"].shape
    steps = np.linspace(min_t, 1.0, num_t)[::-1]
    dt = 1.0 / num_t
    was_training = model.training
    model.eval()
    traj = []
    diffuse_mask = (1 - feats["fixed_mask"]) * feats["res_mask"]
    if self_condition:
        feats["t"] = torch.full((B,), float(steps[0]), device=feats["rigids_t"].device)
        feats["sc_ca_t"] = model(feats)["rigids"][..., 4:]
    out = None
    for i, t in enumerate(steps):
        if t > min_t:
            feats["t"] = torch.full((B,), float(t), device=feats["rigids_t"].device)
            out = model(feats)
            if self_condition:
                feats["sc_ca_t"] = out["rigids"][..., 4:]
            noise = None if noise_fn is None else noise_fn(i, (B, N, 3))
            feats["rigids_t"] = diffuser.reverse_device(feats["rigids_t"], out["rot_score"], out["trans_score"], float(t), dt,
                                                        diffuse_mask=diffuse_mask, center=center, noise_scale=noise_scale,
                                                        noise=noise, generator=generator)
        else:
            out = model(feats)
            feats["rigids_t"] = out["rigids"]
        if return_traj:
            traj.append(feats["rigids_t"].clone())
    if was_training:
        model.train()
    from . import train_step as ts
    atom37, _ = ts.backbone_atoms(feats["rigids_t"], out["psi"])
    res = dict(rigids=feats["rigids_t"], atom37=atom37, psi=out["psi"])
    if return_traj:
        res["rigid_traj"] = traj
    return res
