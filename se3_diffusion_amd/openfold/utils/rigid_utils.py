"""Frame containers crossing the FrameDiff drop-in boundary (subset of the reference's
openfold/utils/rigid_utils.py that the hot path and its callers use -- SURVEY.md 8a M14 / 8b).

``Rotation`` and ``Rigid`` are light tensor wrappers: callers build them with
``Rigid.from_tensor_7`` / hand them to ``SE3Diffuser.reverse`` / read them back with
``to_tensor_7`` / ``get_trans`` / indexing.  The network itself never uses these objects:
frames travel as raw [.., 7] tensors into the HIP kernels.  Unlike the reference,
matrix -> quaternion uses the closed-form branch method (no ``torch.linalg.eigh``, hence no
host sync); quaternion signs are as arbitrary as the reference's eigenvector signs.
"""
from __future__ import annotations

from typing import Any, Optional, Sequence, Tuple

import torch


# ----------------------------------------------------------------------------- algebra
def rot_matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return torch.einsum("...ij,...jk->...ik", a, b)


def rot_vec_mul(r: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return torch.einsum("...ij,...j->...i", r, t)


def quat_to_rot(quat: torch.Tensor) -> torch.Tensor:
    a, b, c, d = torch.unbind(quat, dim=-1)
    aa, bb, cc, dd = a * a, b * b, c * c, d * d
    r0 = torch.stack([aa + bb - cc - dd, 2 * (b * c - a * d), 2 * (b * d + a * c)], dim=-1)
    r1 = torch.stack([2 * (b * c + a * d), aa - bb + cc - dd, 2 * (c * d - a * b)], dim=-1)
    r2 = torch.stack([2 * (b * d - a * c), 2 * (c * d + a * b), aa - bb - cc + dd], dim=-1)
    return torch.stack([r0, r1, r2], dim=-2)


def rot_to_quat(rot: torch.Tensor) -> torch.Tensor:
    """Unit quaternion (w, x, y, z) of a rotation matrix, branch on the largest of (trace, m00, m11, m22)."""
    if rot.shape[-2:] != (3, 3):
        raise ValueError("Input rotation is incorrectly shaped")
    m = rot
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    tr = m00 + m11 + m22
    cand = torch.stack([
        torch.stack([1 + tr, m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]], -1),
        torch.stack([m[..., 2, 1] - m[..., 1, 2], 1 + m00 - m11 - m22, m[..., 0, 1] + m[..., 1, 0], m[..., 0, 2] + m[..., 2, 0]], -1),
        torch.stack([m[..., 0, 2] - m[..., 2, 0], m[..., 0, 1] + m[..., 1, 0], 1 - m00 + m11 - m22, m[..., 1, 2] + m[..., 2, 1]], -1),
        torch.stack([m[..., 1, 0] - m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1], 1 - m00 - m11 + m22], -1),
    ], dim=-2)
    which = torch.argmax(torch.stack([tr, m00, m11, m22], dim=-1), dim=-1)
    q = torch.gather(cand, -2, which[..., None, None].expand(*which.shape, 1, 4)).squeeze(-2)
    return q / torch.linalg.norm(q, dim=-1, keepdim=True)


def quat_multiply(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    a1, b1, c1, d1 = torch.unbind(q1, -1)
    a2, b2, c2, d2 = torch.unbind(q2, -1)
    return torch.stack([
        a1 * a2 - b1 * b2 - c1 * c2 - d1 * d2,
        a1 * b2 + b1 * a2 + c1 * d2 - d1 * c2,
        a1 * c2 - b1 * d2 + c1 * a2 + d1 * b2,
        a1 * d2 + b1 * c2 - c1 * b2 + d1 * a2,
    ], dim=-1)


def quat_multiply_by_vec(quat: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    zero = torch.zeros_like(vec[..., :1])
    return quat_multiply(quat, torch.cat([zero, vec], dim=-1))


def invert_rot_mat(rot_mat: torch.Tensor) -> torch.Tensor:
    return rot_mat.transpose(-1, -2)


def invert_quat(quat: torch.Tensor) -> torch.Tensor:
    conj = quat * quat.new_tensor([1.0, -1.0, -1.0, -1.0])
    return conj / torch.sum(quat ** 2, dim=-1, keepdim=True)


def identity_rot_mats(batch_dims, dtype=None, device=None, requires_grad=True):
    r = torch.eye(3, dtype=dtype, device=device).expand(*batch_dims, 3, 3).clone()
    return r.requires_grad_(requires_grad)


def identity_trans(batch_dims, dtype=None, device=None, requires_grad=True):
    return torch.zeros((*batch_dims, 3), dtype=dtype, device=device, requires_grad=requires_grad)


def identity_quats(batch_dims, dtype=None, device=None, requires_grad=True):
    q = torch.zeros((*batch_dims, 4), dtype=dtype, device=device)
    q[..., 0] = 1
    return q.requires_grad_(requires_grad)


# ----------------------------------------------------------------------------- Rotation
class Rotation:
    """A batch of 3-D rotations held either as [*, 3, 3] matrices or [*, 4] quaternions (fp32)."""

    def __init__(self, rot_mats: Optional[torch.Tensor] = None, quats: Optional[torch.Tensor] = None,
                 normalize_quats: bool = True):
        if (rot_mats is None) == (quats is None):
            raise ValueError("Exactly one input argument must be specified")
        if (rot_mats is not None and rot_mats.shape[-2:] != (3, 3)) or (quats is not None and quats.shape[-1] != 4):
            raise ValueError("Incorrectly shaped rotation matrix or quaternion")
        if quats is not None:
            quats = quats.type(torch.float32)
            if normalize_quats:
                quats = quats / torch.linalg.norm(quats, dim=-1, keepdim=True)
        if rot_mats is not None:
            rot_mats = rot_mats.type(torch.float32)
        self._rot_mats = rot_mats
        self._quats = quats

    @staticmethod
    def identity(shape, dtype=None, device=None, requires_grad=True, fmt="quat"):
        if fmt == "rot_mat":
            return Rotation(rot_mats=identity_rot_mats(shape, dtype, device, requires_grad))
        if fmt == "quat":
            return Rotation(quats=identity_quats(shape, dtype, device, requires_grad), normalize_quats=False)
        raise ValueError(f"Invalid format: f{fmt}")

    def _cur(self):
        return self._rot_mats if self._rot_mats is not None else self._quats

    def __getitem__(self, index: Any) -> "Rotation":
        if type(index) != tuple:
            index = (index,)
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats[index + (slice(None), slice(None))])
        return Rotation(quats=self._quats[index + (slice(None),)], normalize_quats=False)

    def __mul__(self, right: torch.Tensor) -> "Rotation":
        if not isinstance(right, torch.Tensor):
            raise TypeError("The other multiplicand must be a Tensor")
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats * right[..., None, None])
        return Rotation(quats=self._quats * right[..., None], normalize_quats=False)

    __rmul__ = __mul__

    @property
    def shape(self) -> torch.Size:
        return self._rot_mats.shape[:-2] if self._rot_mats is not None else self._quats.shape[:-1]

    @property
    def dtype(self):
        return self._cur().dtype

    @property
    def device(self):
        return self._cur().device

    @property
    def requires_grad(self) -> bool:
        return self._cur().requires_grad

    def get_rot_mats(self) -> torch.Tensor:
        return self._rot_mats if self._rot_mats is not None else quat_to_rot(self._quats)

    def get_quats(self) -> torch.Tensor:
        return self._quats if self._quats is not None else rot_to_quat(self._rot_mats)

    def get_cur_rot(self) -> torch.Tensor:
        return self._cur()

    def get_rotvec(self, eps=1e-6) -> torch.Tensor:
        q = self.get_quats()
        q = torch.where(q[..., :1] < 0, -q, q)
        angle = 2 * torch.atan2(torch.linalg.norm(q[..., 1:], dim=-1), q[..., 0])
        a2 = angle * angle
        small = 2 + a2 / 12 + 7 * a2 * a2 / 2880
        large = angle / torch.sin(angle / 2 + eps)
        scale = torch.where(angle <= 1e-3, small, large)
        return scale[..., None] * q[..., 1:]

    def compose_q_update_vec(self, q_update_vec, normalize_quats=True, update_mask=None) -> "Rotation":
        quats = self.get_quats()
        upd = quat_multiply_by_vec(quats, q_update_vec)
        if update_mask is not None:
            upd = upd * update_mask
        return Rotation(quats=quats + upd, normalize_quats=normalize_quats)

    def compose_r(self, r: "Rotation") -> "Rotation":
        return Rotation(rot_mats=rot_matmul(self.get_rot_mats(), r.get_rot_mats()))

    def compose_q(self, r: "Rotation", normalize_quats=True) -> "Rotation":
        return Rotation(quats=quat_multiply(self.get_quats(), r.get_quats()), normalize_quats=normalize_quats)

    def apply(self, pts: torch.Tensor) -> torch.Tensor:
        return rot_vec_mul(self.get_rot_mats(), pts)

    def invert_apply(self, pts: torch.Tensor) -> torch.Tensor:
        return rot_vec_mul(invert_rot_mat(self.get_rot_mats()), pts)

    def invert(self) -> "Rotation":
        if self._rot_mats is not None:
            return Rotation(rot_mats=invert_rot_mat(self._rot_mats))
        return Rotation(quats=invert_quat(self._quats), normalize_quats=False)

    def unsqueeze(self, dim: int) -> "Rotation":
        if dim >= len(self.shape):
            raise ValueError("Invalid dimension")
        if self._rot_mats is not None:
            return Rotation(rot_mats=self._rot_mats.unsqueeze(dim if dim >= 0 else dim - 2))
        return Rotation(quats=self._quats.unsqueeze(dim if dim >= 0 else dim - 1), normalize_quats=False)

    @staticmethod
    def cat(rs: Sequence["Rotation"], dim: int) -> "Rotation":
        mats = torch.cat([r.get_rot_mats() for r in rs], dim=dim if dim >= 0 else dim - 2)
        return Rotation(rot_mats=mats)

    def map_tensor_fn(self, fn) -> "Rotation":
        if self._rot_mats is not None:
            m = self._rot_mats.view(self._rot_mats.shape[:-2] + (9,))
            m = torch.stack(list(map(fn, torch.unbind(m, dim=-1))), dim=-1)
            return Rotation(rot_mats=m.view(m.shape[:-1] + (3, 3)))
        q = torch.stack(list(map(fn, torch.unbind(self._quats, dim=-1))), dim=-1)
        return Rotation(quats=q, normalize_quats=False)

    def _map(self, f) -> "Rotation":
        if self._rot_mats is not None:
            return Rotation(rot_mats=f(self._rot_mats))
        return Rotation(quats=f(self._quats), normalize_quats=False)

    def cuda(self) -> "Rotation":
        return self._map(lambda x: x.cuda())

    def to(self, device=None, dtype=None) -> "Rotation":
        return self._map(lambda x: x.to(device=device, dtype=dtype))

    def detach(self) -> "Rotation":
        return self._map(lambda x: x.detach())


# ----------------------------------------------------------------------------- Rigid
class Rigid:
    """A batch of rigid transforms: a Rotation plus a [*, 3] translation (fp32)."""

    def __init__(self, rots: Optional[Rotation], trans: Optional[torch.Tensor]):
        if rots is None and trans is None:
            raise ValueError("At least one input argument must be specified")
        if rots is None:
            rots = Rotation.identity(trans.shape[:-1], trans.dtype, trans.device, trans.requires_grad)
        elif trans is None:
            trans = identity_trans(rots.shape, rots.dtype, rots.device, rots.requires_grad)
        if rots.shape != trans.shape[:-1] or rots.device != trans.device:
            raise ValueError("Rots and trans incompatible")
        self._rots = rots
        self._trans = trans.type(torch.float32)

    @staticmethod
    def identity(shape, dtype=None, device=None, requires_grad=True, fmt="quat") -> "Rigid":
        return Rigid(Rotation.identity(shape, dtype, device, requires_grad, fmt=fmt),
                     identity_trans(shape, dtype, device, requires_grad))

    def __getitem__(self, index: Any) -> "Rigid":
        if type(index) != tuple:
            index = (index,)
        return Rigid(self._rots[index], self._trans[index + (slice(None),)])

    def __mul__(self, right: torch.Tensor) -> "Rigid":
        if not isinstance(right, torch.Tensor):
            raise TypeError("The other multiplicand must be a Tensor")
        return Rigid(self._rots * right, self._trans * right[..., None])

    __rmul__ = __mul__

    @property
    def shape(self) -> torch.Size:
        return self._trans.shape[:-1]

    @property
    def device(self):
        return self._trans.device

    def get_rots(self) -> Rotation:
        return self._rots

    def get_trans(self) -> torch.Tensor:
        return self._trans

    def compose_q_update_vec(self, q_update_vec, update_mask=None) -> "Rigid":
        q_vec, t_vec = q_update_vec[..., :3], q_update_vec[..., 3:]
        new_rots = self._rots.compose_q_update_vec(q_vec, update_mask=update_mask)
        upd = self._rots.apply(t_vec)
        if update_mask is not None:
            upd = upd * update_mask
        return Rigid(new_rots, self._trans + upd)

    def compose(self, r: "Rigid") -> "Rigid":
        return Rigid(self._rots.compose_r(r._rots), self._rots.apply(r._trans) + self._trans)

    def compose_r(self, rot: Rotation, order="right") -> "Rigid":
        if order == "right":
            return Rigid(self._rots.compose_r(rot), self._trans)
        if order == "left":
            return Rigid(rot.compose_r(self._rots), self._trans)
        raise ValueError(f"Unrecognized multiplication order: {order}")

    def apply(self, pts: torch.Tensor) -> torch.Tensor:
        return self._rots.apply(pts) + self._trans

    def invert_apply(self, pts: torch.Tensor) -> torch.Tensor:
        return self._rots.invert_apply(pts - self._trans)

    def invert(self) -> "Rigid":
        inv = self._rots.invert()
        return Rigid(inv, -1 * inv.apply(self._trans))

    def map_tensor_fn(self, fn) -> "Rigid":
        t = torch.stack(list(map(fn, torch.unbind(self._trans, dim=-1))), dim=-1)
        return Rigid(self._rots.map_tensor_fn(fn), t)

    def to_tensor_4x4(self) -> torch.Tensor:
        t = self._trans.new_zeros((*self.shape, 4, 4))
        t[..., :3, :3] = self._rots.get_rot_mats()
        t[..., :3, 3] = self._trans
        t[..., 3, 3] = 1
        return t

    @staticmethod
    def from_tensor_4x4(t: torch.Tensor) -> "Rigid":
        if t.shape[-2:] != (4, 4):
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(rot_mats=t[..., :3, :3]), t[..., :3, 3])

    def to_tensor_7(self) -> torch.Tensor:
        return torch.cat([self._rots.get_quats(), self._trans], dim=-1)

    @staticmethod
    def from_tensor_7(t: torch.Tensor, normalize_quats: bool = False) -> "Rigid":
        if t.shape[-1] != 7:
            raise ValueError("Incorrectly shaped input tensor")
        return Rigid(Rotation(quats=t[..., :4], normalize_quats=normalize_quats), t[..., 4:])

    @staticmethod
    def from_3_points(p_neg_x_axis, origin, p_xy_plane, eps: float = 1e-8) -> "Rigid":
        """Gram-Schmidt frames (AF2 alg. 21): x axis along origin - p_neg_x_axis."""
        e0 = origin - p_neg_x_axis
        e1 = p_xy_plane - origin
        e0 = e0 / torch.sqrt((e0 * e0).sum(-1, keepdim=True) + eps)
        e1 = e1 - e0 * (e0 * e1).sum(-1, keepdim=True)
        e1 = e1 / torch.sqrt((e1 * e1).sum(-1, keepdim=True) + eps)
        e2 = torch.cross(e0, e1, dim=-1)
        return Rigid(Rotation(rot_mats=torch.stack([e0, e1, e2], dim=-1)), origin)

    def unsqueeze(self, dim: int) -> "Rigid":
        if dim >= len(self.shape):
            raise ValueError("Invalid dimension")
        return Rigid(self._rots.unsqueeze(dim), self._trans.unsqueeze(dim if dim >= 0 else dim - 1))

    @staticmethod
    def cat(ts: Sequence["Rigid"], dim: int) -> "Rigid":
        return Rigid(Rotation.cat([t._rots for t in ts], dim),
                     torch.cat([t._trans for t in ts], dim=dim if dim >= 0 else dim - 1))

    def apply_rot_fn(self, fn) -> "Rigid":
        return Rigid(fn(self._rots), self._trans)

    def apply_trans_fn(self, fn) -> "Rigid":
        return Rigid(self._rots, fn(self._trans))

    def scale_translation(self, trans_scale_factor: float) -> "Rigid":
        return self.apply_trans_fn(lambda t: t * trans_scale_factor)

    def stop_rot_gradient(self) -> "Rigid":
        return self.apply_rot_fn(lambda r: r.detach())

    def cuda(self) -> "Rigid":
        return Rigid(self._rots.cuda(), self._trans.cuda())

    def to(self, device=None, dtype=None) -> "Rigid":
        return Rigid(self._rots.to(device=device), self._trans.to(device=device))
