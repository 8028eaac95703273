"""Caller-side pieces of one FrameDiff training step, used by bench.py and the tests.

The reference's Experiment.loss_fn (experiments/train_se3_diffusion.py:524-693) and Adam
(:139) are CALLER code that runs unchanged on top of ScoreNetwork.forward; here they are
restated compactly with plain torch ops (same arithmetic, config/base.yaml weights) so the
benchmark step = forward + DSM loss + backward + optimizer, as the reference trains.
Synthetic batches follow SURVEY.md 8d (no PDB data is available offline).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch

from . import hip, trunk

EXP = SimpleNamespace(  # config/base.yaml:104-115
    trans_loss_weight=1.0, rot_loss_weight=0.5, rot_loss_t_threshold=0.2, separate_rot_loss=True,
    trans_x0_threshold=1.0, coordinate_scaling=0.1, bb_atom_loss_weight=1.0, bb_atom_loss_t_filter=0.25,
    dist_mat_loss_weight=1.0, dist_mat_loss_t_filter=0.25, aux_loss_weight=0.25)


def base_model_conf(num_blocks=4):
    """config/base.yaml:45-67 as attribute namespaces (stands in for the OmegaConf node)."""
    ns = SimpleNamespace
    return ns(node_embed_size=256, edge_embed_size=128, dropout=0.0,
              embed=ns(index_embed_size=32, aatype_embed_size=64, embed_self_conditioning=True, num_bins=22,
                       min_bin=1e-5, max_bin=20.0),
              ipa=ns(c_s=256, c_z=128, c_hidden=256, c_skip=64, no_heads=8, no_qk_points=8, no_v_points=12,
                     seq_tfmr_num_heads=4, seq_tfmr_num_layers=2, num_blocks=num_blocks, coordinate_scaling=0.1))


def perturb_final_layers(model, seed=0, scale=0.02):
    """The reference zero-initialises its 'final' layers, so at init rot_score = psi = 0 and half the path carries
    no signal.  Benchmarks and tests perturb them with small noise (all-zero operands would also
    flatter the clocks, cdna guide rule 25)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 2 and float(p.abs().max()) == 0.0:
                p.copy_((torch.randn(p.shape, generator=g) * scale / math.sqrt(p.shape[1])).to(p.device))


def backbone_atoms(rigids, psi):
    """atom37 [.., 37, 3], atom14 from frames (A) + psi -- fd_backbone_atoms (all_atom.py:152-174)."""
    shp = rigids.shape[:-1]
    r = rigids.reshape(-1, 7).to(torch.float32).contiguous()
    p = psi.reshape(-1, 2).to(torch.float32).contiguous()
    R = r.shape[0]
    a37 = torch.empty((R, 37, 3), device=r.device)
    a14 = torch.empty((R, 14, 3), device=r.device)
    hip.get_lib().call("fd_backbone_atoms", r, p, trunk.head_const(), a37, a14, R)
    return a37.view(*shp, 37, 3), a14.view(*shp, 14, 3)


def synthetic_batch(B, N, device, seed=0):
    """Same-length training batch with the keys the reference DataLoader provides
    (pdb_data_loader.py:220-276): random-walk CA trace, random frames, t ~ U(0.01, 1)."""
    rs = np.random.RandomState(seed)

    def frames():
        steps = rs.standard_normal((B, N, 3))
        ca = np.cumsum(3.8 * steps / np.linalg.norm(steps, axis=-1, keepdims=True), 1)
        ca -= ca.mean(1, keepdims=True)
        q = rs.standard_normal((B, N, 4))
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
        return np.concatenate([q, ca], -1)

    r0 = frames()
    t = rs.uniform(0.01, 1.0, size=(B,))
    beta = t * 0.1 + 0.5 * t ** 2 * 19.9
    # noised frames: VP-SDE on translations (r3_diffuser.py:81-101), random rotation perturbation
    x0 = r0[..., 4:] * 0.1
    xt = np.exp(-0.5 * beta)[:, None, None] * x0 + np.sqrt(1 - np.exp(-beta))[:, None, None] * rs.standard_normal(x0.shape)
    trans_score = -(xt - np.exp(-0.5 * beta)[:, None, None] * x0) / (1 - np.exp(-beta))[:, None, None]
    qn = r0[..., :4] + (0.1 + t)[:, None, None] * rs.standard_normal((B, N, 4))
    qn /= np.linalg.norm(qn, axis=-1, keepdims=True)
    rt = np.concatenate([qn, xt * 10.0], -1)
    rot_score = rs.standard_normal((B, N, 3)) * (0.5 / (0.1 + t))[:, None, None]
    tor = rs.standard_normal((B, N, 7, 2))
    tor /= np.linalg.norm(tor, axis=-1, keepdims=True)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
    return dict(
        res_mask=torch.ones(B, N, device=device), fixed_mask=torch.zeros(B, N, device=device),
        seq_idx=torch.arange(1, N + 1, device=device)[None].repeat(B, 1), t=f32(t),
        sc_ca_t=torch.zeros(B, N, 3, device=device), rigids_t=f32(rt), rigids_0=f32(r0),
        torsion_angles_sin_cos=f32(tor), rot_score=f32(rot_score), trans_score=f32(trans_score),
        rot_score_scaling=f32(1.0 / (0.1 + t)), trans_score_scaling=f32(1.0 / np.sqrt(1 - np.exp(-beta))),
    )


def dsm_loss(batch, out, gt_atom37, exp=EXP):
    """Experiment.loss_fn arithmetic (train_se3_diffusion.py:538-666), both rotation-loss branches."""
    bb_mask = batch["res_mask"]
    diffuse_mask = 1 - batch["fixed_mask"]
    loss_mask = bb_mask * diffuse_mask
    B, N = bb_mask.shape
    t = batch["t"]
    denom = loss_mask.sum(dim=-1) + 1e-10
    pred_rot = out["rot_score"] * diffuse_mask[..., None]
    pred_trans = out["trans_score"] * diffuse_mask[..., None]
    trans_score_loss = (((batch["trans_score"] - pred_trans) ** 2 * loss_mask[..., None])
                        / batch["trans_score_scaling"][:, None, None] ** 2).sum(dim=(-1, -2)) / denom
    gt_x0 = batch["rigids_0"][..., 4:] * exp.coordinate_scaling
    pred_x0 = out["rigids"][..., 4:] * exp.coordinate_scaling
    trans_x0_loss = ((gt_x0 - pred_x0) ** 2 * loss_mask[..., None]).sum(dim=(-1, -2)) / denom
    trans_loss = (trans_score_loss * (t > exp.trans_x0_threshold) + trans_x0_loss * (t <= exp.trans_x0_threshold))
    trans_loss = trans_loss * exp.trans_loss_weight
    gt_angle = torch.norm(batch["rot_score"], dim=-1, keepdim=True)
    gt_axis = batch["rot_score"] / (gt_angle + 1e-6)
    pr_angle = torch.norm(pred_rot, dim=-1, keepdim=True)
    pr_axis = pred_rot / (pr_angle + 1e-6)
    axis_loss = ((gt_axis - pr_axis) ** 2 * loss_mask[..., None]).sum(dim=(-1, -2)) / denom
    angle_loss = (((gt_angle - pr_angle) ** 2 * loss_mask[..., None])
                  / batch["rot_score_scaling"][:, None, None] ** 2).sum(dim=(-1, -2)) / denom
    angle_loss = angle_loss * exp.rot_loss_weight * (t > exp.rot_loss_t_threshold)
    rot_loss = angle_loss + axis_loss
    if not getattr(exp, "separate_rot_loss", True):            # :597-604 (config/icml_published.yaml)
        rot_loss = (((batch["rot_score"] - pred_rot) ** 2 * loss_mask[..., None])
                    / batch["rot_score_scaling"][:, None, None] ** 2).sum(dim=(-1, -2)) / denom
        rot_loss = rot_loss * exp.rot_loss_weight * (t > exp.rot_loss_t_threshold)
    pred_atoms = out["atom37"][:, :, :5]
    gt_atoms = gt_atom37[:, :, :5]
    atom_mask = torch.any(gt_atoms != 0, dim=-1).to(pred_atoms.dtype) * loss_mask[..., None]
    bb_atom_loss = ((pred_atoms - gt_atoms) ** 2 * atom_mask[..., None]).sum(dim=(-1, -2, -3)) / (atom_mask.sum(dim=(-1, -2)) + 1e-10)
    bb_atom_loss = bb_atom_loss * exp.bb_atom_loss_weight * (t < exp.bb_atom_loss_t_filter) * exp.aux_loss_weight
    gflat = gt_atoms.reshape(B, N * 5, 3)
    pflat = pred_atoms.reshape(B, N * 5, 3)
    gd = torch.linalg.norm(gflat[:, :, None, :] - gflat[:, None, :, :], dim=-1)
    pd = torch.linalg.norm(pflat[:, :, None, :] - pflat[:, None, :, :] + 0.0, dim=-1)
    flm = loss_mask[:, :, None].expand(B, N, 5).reshape(B, N * 5)
    frm = bb_mask[:, :, None].expand(B, N, 5).reshape(B, N * 5)
    gd = gd * flm[..., None]
    pd = pd * flm[..., None]
    pmask = flm[..., None] * frm[:, None, :] * (gd < 6)
    dist_loss = ((gd - pd) ** 2 * pmask).sum(dim=(1, 2)) / (pmask.sum(dim=(1, 2)) - N)
    dist_loss = dist_loss * exp.dist_mat_loss_weight * (t < exp.dist_mat_loss_t_filter) * exp.aux_loss_weight
    final = rot_loss + trans_loss + bb_atom_loss + dist_loss
    return final.sum() / (torch.any(bb_mask > 0, dim=-1).sum() + 1e-10)
