"""Stand-alone, differentiable score-head entry points backed by the fd_heads kernels.

Inside ScoreNetwork.forward the heads are already fused; these wrappers exist so that the
reference's public diffuser API -- SE3Diffuser.calc_rot_score / calc_trans_score and
SO3Diffuser.torch_score (se3_diffuser.py:115-125, so3_diffuser.py:274-305) -- runs on the
same HIP kernels when a caller uses it directly on GPU tensors."""
from __future__ import annotations

import torch

from . import hip, trunk


def _dconf(se3=None, so3=None, r3=None):
    so3 = so3 if so3 is not None else getattr(se3, "_so3_diffuser", None)
    r3 = r3 if r3 is not None else getattr(se3, "_r3_diffuser", None)
    cs = float(getattr(getattr(r3, "_r3_conf", None), "coordinate_scaling", 0.1))
    return (cs, float(getattr(r3, "min_b", 0.1)), float(getattr(r3, "max_b", 20.0)),
            float(getattr(so3, "min_sigma", 0.1)), float(getattr(so3, "max_sigma", 1.5)), 1000,
            so3 if getattr(so3, "use_cached_score", False) else None, int(getattr(so3, "num_sigma", 1000)))


class _RotScoreFn(torch.autograd.Function):
    """rot_score[B,N,3] (float64) of q_0t = inv(q_0) (x) q_t; differentiable w.r.t. q_0."""

    @staticmethod
    def forward(ctx, quat_t, quat_0, t, dconf):
        B, N = quat_t.shape[:2]
        dev = quat_t
        R = B * N
        rig0 = torch.zeros((R, 7), device=dev.device)
        rig0[:, :4] = quat_t.reshape(R, 4)
        qf = quat_0.reshape(R, 4).to(torch.float32).contiguous()
        z3 = torch.zeros((R, 3), device=dev.device)
        u = torch.ones((R, 2), device=dev.device)
        gt = torch.zeros((R, 14), device=dev.device)
        ones = torch.ones((R,), device=dev.device)
        zeros = torch.zeros((R,), device=dev.device)
        tt = t.to(torch.float32).reshape(-1).contiguous()
        if tt.numel() == 1 and B > 1:
            tt = tt.expand(B).contiguous()
        hc, sg = trunk.head_tables(dconf, dev.device)
        rot = torch.empty((B, N, 3), device=dev.device, dtype=torch.float64)
        junk = [torch.empty((R, k), device=dev.device) for k in (3, 7, 2, 111, 42)]
        hip.get_lib().call("fd_heads_fwd", rig0, qf, z3, u, (gt, 4), 14, zeros, ones, tt, sg, sg.numel(), hc,
                           rot, junk[0], junk[1], junk[2], junk[3], junk[4], None, B, N)
        ctx.saved = (rig0, qf, z3, u, junk[2], zeros, ones, tt, sg, hc, B, N)
        return rot

    @staticmethod
    def backward(ctx, d_rot):
        rig0, qf, z3, u, psi, zeros, ones, tt, sg, hc, B, N = ctx.saved
        R = B * N
        dq = torch.empty((R, 4), device=qf.device)
        dt = torch.empty((R, 3), device=qf.device)
        du = torch.empty((R, 2), device=qf.device)
        hip.get_lib().call("fd_heads_bwd", rig0, qf, z3, u, psi, zeros, ones, tt, sg, sg.numel(), hc,
                           d_rot.to(torch.float64).contiguous(), None, None, None, None, dq, dt, du, B, N)
        return None, dq.view(B, N, 4), None, None


def rot_score(quat_t, quat_0, t, se3=None):
    """SE3Diffuser.calc_rot_score on quaternion tensors [B,N,4]."""
    return _RotScoreFn.apply(quat_t.to(torch.float32).contiguous(), quat_0, t, _dconf(se3))


def rotvec_score(vec, t, so3):
    """SO3Diffuser.torch_score on GPU rotation vectors [B,N,3] (exp map -> quaternion -> fused kernel)."""
    squeeze = vec.dim() == 2
    v = vec[None] if squeeze else vec
    ang2 = (v * v).sum(-1, keepdim=True)
    ang = torch.sqrt(ang2)
    small = ang <= 1e-3
    sc = torch.where(small, 0.5 - ang2 / 48 + ang2 * ang2 / 3840, torch.sin(ang / 2) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(ang / 2), sc * v], -1).to(torch.float32)
    ident = torch.zeros_like(q)
    ident[..., 0] = 1
    conj = q * q.new_tensor([1.0, -1.0, -1.0, -1.0])
    out = _RotScoreFn.apply(ident, conj, t, _dconf(so3=so3))
    return out[0] if squeeze else out


def trans_score(x_t, x_0, t, r3, scale=True):
    """R3Diffuser.score with use_torch=True (elementwise; plain torch ops on whatever device the inputs live)."""
    return r3.score(x_t, x_0, t, use_torch=True, scale=scale)
