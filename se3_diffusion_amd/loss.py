"""Fused DSM training loss (fd_dsm_loss): the arithmetic of ``Experiment.loss_fn``
(experiments/train_se3_diffusion.py:524-693, both ``separate_rot_loss`` branches) -- value and gradient w.r.t. the network
outputs in three HIP launches instead of ~150 torch kernels over materialised [B,5N,5N] tensors.

    loss, aux = dsm_loss(batch, model_out, gt_atom37)        # same arguments as train_step.dsm_loss
    loss.backward()

``aux`` carries the per-example terms the reference logs.  train_step.dsm_loss (plain torch, the same formulas) is the
checker in tests/test_loss.py.
"""
from __future__ import annotations

import torch

from . import hip
from .train_step import EXP


class _DsmLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rot_score, trans_score, rigids, atom37, batch, gt_atom37, exp):
        lib = hip.get_lib()
        dev = trans_score.device
        B, N = batch["res_mask"].shape
        f32 = lambda x: x.detach().to(device=dev, dtype=torch.float32).contiguous()
        f64 = lambda x: x.detach().to(device=dev, dtype=torch.float64).contiguous()
        keep = dict(
            res_mask=f32(batch["res_mask"]), fixed_mask=f32(batch["fixed_mask"]), t=f32(batch["t"]),
            gt_trans_score=f32(batch["trans_score"]), gt_rot_score=f64(batch["rot_score"]),
            trans_score_scaling=f32(batch["trans_score_scaling"]), rot_score_scaling=f32(batch["rot_score_scaling"]),
            gt_rigids=f32(batch["rigids_0"]), gt_atom37=f32(gt_atom37),
            rot_score=f64(rot_score), trans_score=f32(trans_score), rigids=f32(rigids), atom37=f32(atom37))
        out = dict(
            d_rot_score=torch.empty(B, N, 3, device=dev, dtype=torch.float64),
            d_trans_score=torch.empty(B, N, 3, device=dev), d_rigids=torch.empty(B, N, 7, device=dev),
            d_atom37=torch.empty(B, N, 37, 3, device=dev), terms=torch.empty(B, 8, device=dev),
            loss=torch.empty(1, device=dev), scratch=torch.empty(B * N * 15 + 2 * B, device=dev))
        d = hip.FdLossDesc()
        d.B, d.N = B, N
        for k, v in {**keep, **out}.items():
            setattr(d, k, v.data_ptr())
        for k in ("coordinate_scaling", "trans_x0_threshold", "trans_loss_weight", "rot_loss_weight",
                  "rot_loss_t_threshold", "bb_atom_loss_weight", "bb_atom_loss_t_filter", "aux_loss_weight",
                  "dist_mat_loss_weight", "dist_mat_loss_t_filter"):
            setattr(d, k, float(getattr(exp, k)))
        d.joint_rot_loss = 0 if getattr(exp, "separate_rot_loss", True) else 1
        if lib.is_device and not trans_score.is_cuda:
            raise hip.FdError("fd_dsm_loss: the outputs are not on the GPU; the hot path has no CPU fallback")
        lib.call("fd_dsm_loss", d)
        ctx.grads = (out["d_rot_score"].to(rot_score.dtype), out["d_trans_score"].to(trans_score.dtype),
                     out["d_rigids"].to(rigids.dtype), out["d_atom37"].to(atom37.dtype))
        ctx.mark_non_differentiable(out["terms"])
        return out["loss"][0], out["terms"]

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        grads = ctx.grads
        ctx.grads = None
        return tuple(g * g_loss for g in grads) + (None, None, None)


def dsm_loss(batch, out, gt_atom37, exp=EXP, with_terms=False):
    """Same arguments and value as ``train_step.dsm_loss``; differentiable w.r.t. the four network outputs."""
    loss, terms = _DsmLossFn.apply(out["rot_score"], out["trans_score"], out["rigids"], out["atom37"], batch,
                                   gt_atom37, exp)
    if with_terms:
        # separate_rot_loss=False: axis_loss is 0 and the angle_loss slot carries the joint rot-score MSE
        names = ("trans_score_loss", "trans_x0_loss", "axis_loss", "angle_loss", "bb_atom_loss", "dist_mat_loss",
                 "final", "loss_mask_sum")
        return loss, {n: terms[:, i] for i, n in enumerate(names)}
    return loss
