"""Hot-path subset of the reference's data/utils.py: array helpers the
diffuser and the experiment scripts import as ``du``.  scipy's Rotation conversions
(reference data/utils.py:184-198) are restated in numpy on unit quaternions."""
import numpy as np
import torch

move_to_np = lambda x: x.cpu().detach().numpy()  # noqa: E731


# ---- numpy rotation-vector algebra (float64) -------------------------------------------------
def rotvec_to_quat_wxyz(rotvec):
    """Rotation vector -> unit quaternion (w, x, y, z); Taylor branch for |v| <= 1e-3 like scipy."""
    rv = np.asarray(rotvec, dtype=np.float64)
    a2 = np.sum(rv * rv, axis=-1)
    ang = np.sqrt(a2)
    small = ang <= 1e-3
    scale = np.where(small, 0.5 - a2 / 48 + a2 * a2 / 3840, np.sin(ang / 2) / np.where(small, 1.0, ang))
    return np.concatenate([np.cos(ang / 2)[..., None], scale[..., None] * rv], axis=-1)


def quat_wxyz_to_rotvec(q):
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    q = np.where(q[..., :1] < 0, -q, q)
    ang = 2 * np.arctan2(np.linalg.norm(q[..., 1:], axis=-1), q[..., 0])
    small = ang <= 1e-3
    a2 = ang * ang
    scale = np.where(small, 2 + a2 / 12 + 7 * a2 * a2 / 2880, ang / np.where(small, 1.0, np.sin(ang / 2)))
    return scale[..., None] * q[..., 1:]


def quat_mul_wxyz(p, q):
    pw, px, py, pz = np.moveaxis(p, -1, 0)
    qw, qx, qy, qz = np.moveaxis(q, -1, 0)
    return np.stack([pw * qw - px * qx - py * qy - pz * qz, pw * qx + px * qw + py * qz - pz * qy,
                     pw * qy - px * qz + py * qw + pz * qx, pw * qz + px * qy - py * qx + pz * qw], axis=-1)


def quat_wxyz_to_matrix(q):
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = np.moveaxis(q, -1, 0)
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def matrix_to_quat_wxyz(m):
    """Branch on the largest of (trace, m00, m11, m22); returns w >= 0."""
    m = np.asarray(m, dtype=np.float64)
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    tr = m00 + m11 + m22
    cand = np.stack([
        np.stack([1 + tr, m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]], -1),
        np.stack([m[..., 2, 1] - m[..., 1, 2], 1 + m00 - m11 - m22, m[..., 0, 1] + m[..., 1, 0], m[..., 0, 2] + m[..., 2, 0]], -1),
        np.stack([m[..., 0, 2] - m[..., 2, 0], m[..., 0, 1] + m[..., 1, 0], 1 - m00 + m11 - m22, m[..., 1, 2] + m[..., 2, 1]], -1),
        np.stack([m[..., 1, 0] - m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1], 1 - m00 - m11 + m22], -1),
    ], -2)
    which = np.argmax(np.stack([tr, m00, m11, m22], -1), -1)
    q = np.take_along_axis(cand, which[..., None, None], axis=-2)[..., 0, :]
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    return np.where(q[..., :1] < 0, -q, q)


def rotvec_to_matrix(rotvec):
    return quat_wxyz_to_matrix(rotvec_to_quat_wxyz(rotvec))


def matrix_to_rotvec(mat):
    return quat_wxyz_to_rotvec(matrix_to_quat_wxyz(mat))


def rotvec_to_quat(rotvec):
    """scipy convention: scalar-last (x, y, z, w)."""
    q = rotvec_to_quat_wxyz(rotvec)
    return np.concatenate([q[..., 1:], q[..., :1]], axis=-1)


def compose_rotvec(r1, r2):
    """R(r1) @ R(r2) as a rotation vector."""
    return quat_wxyz_to_rotvec(quat_mul_wxyz(rotvec_to_quat_wxyz(r1), rotvec_to_quat_wxyz(r2)))


# ---- torch helpers used by callers ------------------------------------------------------------
def calc_distogram(pos, min_bin, max_bin, num_bins):
    d = torch.linalg.norm(pos[:, :, None, :] - pos[:, None, :, :], dim=-1)[..., None]
    lower = torch.linspace(min_bin, max_bin, num_bins, device=pos.device)
    upper = torch.cat([lower[1:], lower.new_tensor([1e8])], dim=-1)
    return ((d > lower) * (d < upper)).type(pos.dtype)


def quat_to_rotvec(quat, eps=1e-6):
    q = torch.where(quat[..., :1] < 0, -quat, quat)
    angle = 2 * torch.atan2(torch.linalg.norm(q[..., 1:], dim=-1), q[..., 0])
    a2 = angle * angle
    scale = torch.where(angle <= 1e-3, 2 + a2 / 12 + 7 * a2 * a2 / 2880, angle / torch.sin(angle / 2 + eps))
    return scale[..., None] * q[..., 1:]
