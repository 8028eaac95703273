"""TEST INFRASTRUCTURE -- CPU restatement of the FrameDiff hot path (the oracle).

A from-scratch, functional torch-CPU restatement of the reference algorithm
(jasonkyuyim/se3_diffusion @ /root/reference) for the path BASELINE.json names:
ScoreNetwork.forward (embedder + IPA trunk + score heads + backbone atoms) and the
SE(3) diffuser arithmetic.  Every function cites the reference file:line it follows.

Pinning: the reference has NO tests or golden vectors, so this
oracle is pinned against outputs of the reference itself, imported in the build
container by oracle/make_golden.py, which also writes tests/golden/*.npz.
tests/test_oracle_golden.py re-checks the oracle against those fixtures everywhere.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module -- as the checker / timed CPU baseline, never as the product path.

Parameters are a flat dict keyed exactly like the reference state_dict.
`dtype` float32 reproduces the reference arithmetic; float64 gives a tight reference
for gradient checks.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# configuration (config/base.yaml:25-67)
# ----------------------------------------------------------------------------
CONF = dict(
    c_s=256, c_z=128, c_hidden=256, c_skip=64, no_heads=8, no_qk_points=8, no_v_points=12,
    tfmr_heads=4, tfmr_layers=2, num_blocks=4, coordinate_scaling=0.1,
    index_embed_size=32, num_bins=22, min_bin=1e-5, max_bin=20.0,
    min_b=0.1, max_b=20.0, min_sigma=0.1, max_sigma=1.5, num_sigma=1000, num_omega=1000,
)


# ----------------------------------------------------------------------------
# quaternion / frame algebra  (openfold/utils/rigid_utils.py)
# ----------------------------------------------------------------------------
def quat_to_rot(q):
    """rigid_utils.py:173-205 (_QTR_MAT contraction written out)."""
    a, b, c, d = q.unbind(-1)
    rows = [
        torch.stack([a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)], -1),
        torch.stack([2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)], -1),
        torch.stack([2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], -1),
    ]
    return torch.stack(rows, -2)


def quat_multiply(p, q):
    """rigid_utils.py:230-263 Hamilton product."""
    pw, px, py, pz = p.unbind(-1)
    qw, qx, qy, qz = q.unbind(-1)
    return torch.stack([
        pw * qw - px * qx - py * qy - pz * qz,
        pw * qx + px * qw + py * qz - pz * qy,
        pw * qy - px * qz + py * qw + pz * qx,
        pw * qz + px * qy - py * qx + pz * qw,
    ], -1)


def quat_multiply_by_vec(q, v):
    """rigid_utils.py:266-275: q (x) (0, v)."""
    w, x, y, z = q.unbind(-1)
    vx, vy, vz = v.unbind(-1)
    return torch.stack([
        -x * vx - y * vy - z * vz,
        w * vx + y * vz - z * vy,
        w * vy - x * vz + z * vx,
        w * vz + x * vy - y * vx,
    ], -1)


def invert_quat(q):
    """rigid_utils.py:282-286."""
    conj = q * q.new_tensor([1.0, -1.0, -1.0, -1.0])
    return conj / (q * q).sum(-1, keepdim=True)


def rot_apply(R, p):
    """rigid_utils.py:82-106 rot_vec_mul."""
    return (R * p[..., None, :]).sum(-1)


def quat_to_rotvec(q, eps=1e-6):
    """data/utils.py:582-599."""
    flip = (q[..., :1] < 0).to(q.dtype)
    q = q * (1 - 2 * flip)
    angle = 2 * torch.atan2(torch.linalg.norm(q[..., 1:], dim=-1), q[..., 0])
    a2 = angle * angle
    small = 2 + a2 / 12 + 7 * a2 * a2 / 2880
    large = angle / torch.sin(angle / 2 + eps)
    is_small = (angle <= 1e-3).to(q.dtype)
    scale = small * is_small + (1 - is_small) * large
    return scale[..., None] * q[..., 1:]


def rot_to_quat(R):
    """rigid_utils.py:208-227 (largest-eigenvalue eigenvector of the 4x4 K matrix).
    Sign is arbitrary (eigh)."""
    xx, xy, xz = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    yx, yy, yz = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    zx, zy, zz = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    k = torch.stack([
        torch.stack([xx + yy + zz, zy - yz, xz - zx, yx - xy], -1),
        torch.stack([zy - yz, xx - yy - zz, xy + yx, xz + zx], -1),
        torch.stack([xz - zx, xy + yx, yy - xx - zz, yz + zy], -1),
        torch.stack([yx - xy, xz + zx, yz + zy, zz - xx - yy], -1),
    ], -2) / 3.0
    _, vec = torch.linalg.eigh(k)
    return vec[..., -1]


# numpy rotation-vector helpers (scipy.spatial.transform.Rotation restated; scipy 1.7.3
# pinned by the reference, se3.yml:237; call sites se3_diffuser.py:16,24, utils.py:192-198)
def rotvec_to_matrix_np(rv):
    rv = np.asarray(rv, dtype=np.float64)
    ang = np.linalg.norm(rv, axis=-1)
    small = ang <= 1e-3
    a2 = ang * ang
    # scipy converts rotvec -> quat with a Taylor branch, then quat -> matrix
    scale = np.where(small, 0.5 - a2 / 48 + a2 * a2 / 3840, np.sin(ang / 2) / np.where(small, 1.0, ang))
    q = np.concatenate([np.cos(ang / 2)[..., None], scale[..., None] * rv], -1)  # (w,x,y,z)
    return quat_to_matrix_np(q)


def quat_to_matrix_np(q):
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1),
    ], -2)
    return m


def matrix_to_quat_np(m):
    """Shepperd's method (what scipy from_matrix does); returns (w,x,y,z) with w >= 0."""
    m = np.asarray(m, dtype=np.float64)
    shp = m.shape[:-2]
    m = m.reshape(-1, 3, 3)
    dec = np.empty((m.shape[0], 4))
    dec[:, :3] = m.diagonal(axis1=1, axis2=2)
    dec[:, 3] = dec[:, :3].sum(1)
    ch = dec.argmax(1)
    q = np.empty((m.shape[0], 4))  # x,y,z,w
    for n in range(m.shape[0]):
        i = ch[n]
        if i != 3:
            j, k = (i + 1) % 3, (i + 2) % 3
            q[n, i] = 1 - dec[n, 3] + 2 * m[n, i, i]
            q[n, j] = m[n, j, i] + m[n, i, j]
            q[n, k] = m[n, k, i] + m[n, i, k]
            q[n, 3] = m[n, k, j] - m[n, j, k]
        else:
            q[n, 0] = m[n, 2, 1] - m[n, 1, 2]
            q[n, 1] = m[n, 0, 2] - m[n, 2, 0]
            q[n, 2] = m[n, 1, 0] - m[n, 0, 1]
            q[n, 3] = 1 + dec[n, 3]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = np.concatenate([q[:, 3:], q[:, :3]], 1)
    q = np.where(q[:, :1] < 0, -q, q)
    return q.reshape(shp + (4,))


def quat_to_rotvec_np(q):
    q = np.where(q[..., :1] < 0, -q, q)
    ang = 2 * np.arctan2(np.linalg.norm(q[..., 1:], axis=-1), q[..., 0])
    small = ang <= 1e-3
    a2 = ang * ang
    scale = np.where(small, 2 + a2 / 12 + 7 * a2 * a2 / 2880, ang / np.where(small, 1.0, np.sin(ang / 2)))
    return scale[..., None] * q[..., 1:]


def matrix_to_rotvec_np(m):
    return quat_to_rotvec_np(matrix_to_quat_np(m))


def compose_rotvec_np(r1, r2):
    """data/utils.py:184-189: R(r1) @ R(r2) -> rotvec."""
    return matrix_to_rotvec_np(rotvec_to_matrix_np(r1) @ rotvec_to_matrix_np(r2))


# ----------------------------------------------------------------------------
# embedder  (model/score_network.py:14-154, data/utils.py:570-580)
# ----------------------------------------------------------------------------
def timestep_embedding(t, dim=32, max_positions=10000):
    """score_network.py:35-47."""
    half = dim // 2
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(max_positions) / (half - 1)))
    emb = (t * max_positions).float()[:, None] * freqs[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


def index_embedding(idx, dim=32, max_len=2056):
    """score_network.py:14-32."""
    k = torch.arange(dim // 2)
    ang = idx[..., None] * math.pi / (max_len ** (2 * k[None] / dim))
    return torch.cat([torch.sin(ang), torch.cos(ang)], -1)


def distogram(pos, min_bin, max_bin, num_bins):
    """data/utils.py:570-580."""
    d = torch.linalg.norm(pos[:, :, None, :] - pos[:, None, :, :], dim=-1)[..., None]
    lower = torch.linspace(min_bin, max_bin, num_bins)
    upper = torch.cat([lower[1:], lower.new_tensor([1e8])])
    return ((d > lower) * (d < upper)).to(pos.dtype)


def node_features(seq_idx, t, fixed_mask, conf=CONF):
    """score_network.py:125-136: [t-emb(32) | fixed(1) | idx-emb(32)] -> 65."""
    B, N = seq_idx.shape
    te = timestep_embedding(t, conf["index_embed_size"])[:, None, :].expand(B, N, -1)
    pt = torch.cat([te, fixed_mask[..., None].float()], -1)
    return torch.cat([pt, index_embedding(seq_idx, conf["index_embed_size"])], -1).float(), pt


def edge_features(seq_idx, t, fixed_mask, sc_ca, conf=CONF):
    """score_network.py:97-101,129-148: [pt_i(33) | pt_j(33) | relidx(32) | dgram(22)] -> 120."""
    B, N = seq_idx.shape
    _, pt = node_features(seq_idx, t, fixed_mask, conf)
    cross = torch.cat([pt[:, :, None, :].expand(B, N, N, -1), pt[:, None, :, :].expand(B, N, N, -1)], -1)
    rel = seq_idx[:, :, None] - seq_idx[:, None, :]
    feats = [cross.float(), index_embedding(rel, conf["index_embed_size"]).float()]
    feats.append(distogram(sc_ca.float(), conf["min_bin"], conf["max_bin"], conf["num_bins"]))
    return torch.cat(feats, -1)


def _mlp3_ln(P, pre, x):
    """nn.Sequential(Linear, ReLU, Linear, ReLU, Linear, LayerNorm) score_network.py:67-86."""
    x = F.relu(F.linear(x, P[f"{pre}.0.weight"], P[f"{pre}.0.bias"]))
    x = F.relu(F.linear(x, P[f"{pre}.2.weight"], P[f"{pre}.2.bias"]))
    x = F.linear(x, P[f"{pre}.4.weight"], P[f"{pre}.4.bias"])
    return F.layer_norm(x, x.shape[-1:], P[f"{pre}.5.weight"], P[f"{pre}.5.bias"])


def embedder(P, seq_idx, t, fixed_mask, sc_ca, conf=CONF, dtype=torch.float32):
    nf, _ = node_features(seq_idx, t, fixed_mask, conf)
    ef = edge_features(seq_idx, t, fixed_mask, sc_ca, conf)
    node = _mlp3_ln(P, "embedding_layer.node_embedder", nf.to(dtype))
    edge = _mlp3_ln(P, "embedding_layer.edge_embedder", ef.to(dtype))
    return node, edge


# ----------------------------------------------------------------------------
# IPA  (model/ipa_pytorch.py:236-471)
# ----------------------------------------------------------------------------
def ipa_projections(P, pre, s, quat, trans, conf=CONF):
    """ipa_pytorch.py:334-374: q,k,v and global-frame q/k/v points."""
    H, C, Pq, Pv = conf["no_heads"], conf["c_hidden"], conf["no_qk_points"], conf["no_v_points"]
    B, N, _ = s.shape
    q = F.linear(s, P[f"{pre}.linear_q.weight"], P[f"{pre}.linear_q.bias"]).view(B, N, H, C)
    kv = F.linear(s, P[f"{pre}.linear_kv.weight"], P[f"{pre}.linear_kv.bias"]).view(B, N, H, 2 * C)
    k, v = kv[..., :C], kv[..., C:]
    R = quat_to_rot(quat)

    def pts(raw, npts):
        x, y, z = raw.split(raw.shape[-1] // 3, dim=-1)       # [x(H*P) | y | z] :351-352
        p = torch.stack([x, y, z], -1)                        # [B,N,H*P,3]
        p = rot_apply(R[:, :, None], p) + trans[:, :, None]   # Rigid.apply rigid_utils.py:1104
        return p.view(B, N, H, npts, 3)

    q_pts = pts(F.linear(s, P[f"{pre}.linear_q_points.weight"], P[f"{pre}.linear_q_points.bias"]), Pq)
    kv_pts = pts(F.linear(s, P[f"{pre}.linear_kv_points.weight"], P[f"{pre}.linear_kv_points.bias"]), Pq + Pv)
    return q, k, v, q_pts, kv_pts[..., :Pq, :], kv_pts[..., Pq:, :]


def ipa(P, pre, s, z, quat, trans, mask, conf=CONF, return_aux=False):
    """ipa_pytorch.py:303-471.  quat/trans: frames in nm (already scaled)."""
    H, C, Pq, Pv = conf["no_heads"], conf["c_hidden"], conf["no_qk_points"], conf["no_v_points"]
    B, N, _ = s.shape
    q, k, v, q_pts, k_pts, v_pts = ipa_projections(P, pre, s, quat, trans, conf)
    b = F.linear(z, P[f"{pre}.linear_b.weight"], P[f"{pre}.linear_b.bias"])          # [B,N,N,H]
    a = torch.einsum("bihc,bjhc->bhij", q, k) * math.sqrt(1.0 / (3 * C))
    a = a + math.sqrt(1.0 / 3) * b.permute(0, 3, 1, 2)
    d2 = ((q_pts[:, :, None] - k_pts[:, None, :]) ** 2).sum(-1)                      # [B,N,N,H,Pq]
    hw = F.softplus(P[f"{pre}.head_weights"]) * math.sqrt(1.0 / (3 * (Pq * 9.0 / 2)))
    pt_att = (d2 * hw[:, None]).sum(-1) * (-0.5)                                     # [B,N,N,H]
    a = a + pt_att.permute(0, 3, 1, 2)
    sq_mask = 1e5 * (mask[:, :, None] * mask[:, None, :] - 1)
    a = torch.softmax(a + sq_mask[:, None], dim=-1)                                  # [B,H,N,N]
    o = torch.einsum("bhij,bjhc->bihc", a, v).reshape(B, N, H * C)
    o_pt = torch.einsum("bhij,bjhpx->bihpx", a, v_pts)                               # global frame
    R = quat_to_rot(quat)
    o_pt = rot_apply(R.transpose(-1, -2)[:, :, None, None], o_pt - trans[:, :, None, None])  # invert_apply :442
    o_norm = torch.sqrt((o_pt ** 2).sum(-1) + 1e-8).reshape(B, N, H * Pv)
    o_pt = o_pt.reshape(B, N, H * Pv, 3)
    pair_z = F.linear(z, P[f"{pre}.down_z.weight"], P[f"{pre}.down_z.bias"])         # [B,N,N,32]
    o_pair = torch.einsum("bhij,bijc->bihc", a, pair_z).reshape(B, N, -1)
    feats = torch.cat([o, o_pt[..., 0], o_pt[..., 1], o_pt[..., 2], o_norm, o_pair], -1)
    out = F.linear(feats, P[f"{pre}.linear_out.weight"], P[f"{pre}.linear_out.bias"])
    if return_aux:
        return out, dict(q=q, k=k, v=v, q_pts=q_pts, k_pts=k_pts, v_pts=v_pts, attn=a, feats=feats)
    return out


# ----------------------------------------------------------------------------
# sequence transformer (torch.nn.TransformerEncoder, post-norm, relu, dropout 0;
# ipa_pytorch.py:584-595,633-638)
# ----------------------------------------------------------------------------
def tfmr_layer(P, pre, x, key_add, nheads):
    """One post-norm encoder layer. key_add [B,N]: additive logit term per key
    (float src_key_padding_mask semantics: +1.0 on padded keys in train/grad mode)."""
    B, N, D = x.shape
    hd = D // nheads
    qkv = F.linear(x, P[f"{pre}.self_attn.in_proj_weight"], P[f"{pre}.self_attn.in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, N, nheads, hd).transpose(1, 2)
    k = k.view(B, N, nheads, hd).transpose(1, 2)
    v = v.view(B, N, nheads, hd).transpose(1, 2)
    a = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + key_add[:, None, None, :]
    a = torch.softmax(a, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, D)
    o = F.linear(o, P[f"{pre}.self_attn.out_proj.weight"], P[f"{pre}.self_attn.out_proj.bias"])
    x = F.layer_norm(x + o, (D,), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"])
    ff = F.linear(F.relu(F.linear(x, P[f"{pre}.linear1.weight"], P[f"{pre}.linear1.bias"])),
                  P[f"{pre}.linear2.weight"], P[f"{pre}.linear2.bias"])
    return F.layer_norm(x + ff, (D,), P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"])


def node_transition(P, pre, s):
    """ipa_pytorch.py:169-191."""
    h = F.relu(F.linear(s, P[f"{pre}.linear_1.weight"], P[f"{pre}.linear_1.bias"]))
    h = F.relu(F.linear(h, P[f"{pre}.linear_2.weight"], P[f"{pre}.linear_2.bias"]))
    h = F.linear(h, P[f"{pre}.linear_3.weight"], P[f"{pre}.linear_3.bias"])
    return F.layer_norm(h + s, s.shape[-1:], P[f"{pre}.ln.weight"], P[f"{pre}.ln.bias"])


def edge_transition(P, pre, node, edge):
    """ipa_pytorch.py:194-233."""
    B, N, _ = node.shape
    e = F.linear(node, P[f"{pre}.initial_embed.weight"], P[f"{pre}.initial_embed.bias"])
    x = torch.cat([edge, e[:, :, None, :].expand(B, N, N, -1), e[:, None, :, :].expand(B, N, N, -1)], -1)
    h = F.relu(F.linear(x, P[f"{pre}.trunk.0.weight"], P[f"{pre}.trunk.0.bias"]))
    h = F.relu(F.linear(h, P[f"{pre}.trunk.2.weight"], P[f"{pre}.trunk.2.bias"]))
    y = F.linear(h + x, P[f"{pre}.final_layer.weight"], P[f"{pre}.final_layer.bias"])
    return F.layer_norm(y, y.shape[-1:], P[f"{pre}.layer_norm.weight"], P[f"{pre}.layer_norm.bias"])


def backbone_update(quat, trans, upd, dmask):
    """rigid_utils.py:587-616,1039-1063: q' = normalize(q + d*(q (x) (0,u_q))),
    t' = t + d * R(q_old) u_t."""
    dq = quat_multiply_by_vec(quat, upd[..., :3]) * dmask[..., None]
    nq = quat + dq
    nq = nq / torch.linalg.norm(nq, dim=-1, keepdim=True)
    nt = trans + rot_apply(quat_to_rot(quat), upd[..., 3:]) * dmask[..., None]
    return nq, nt


# ----------------------------------------------------------------------------
# IGSO(3) / R^3 schedules and scores (data/so3_diffuser.py, data/r3_diffuser.py)
# ----------------------------------------------------------------------------
def so3_sigma(t, conf=CONF):
    """so3_diffuser.py:192-199 (logarithmic schedule)."""
    t = np.asarray(t, dtype=np.float64)
    return np.log(t * np.exp(conf["max_sigma"]) + (1 - t) * np.exp(conf["min_sigma"]))


def so3_discrete_sigma(conf=CONF):
    """so3_diffuser.py:182-186."""
    return so3_sigma(np.linspace(0.0, 1.0, conf["num_sigma"]), conf)


def so3_t_to_idx(t, conf=CONF):
    """so3_diffuser.py:188-190,211-213."""
    return np.digitize(so3_sigma(t, conf), so3_discrete_sigma(conf)) - 1


def so3_diffusion_coef(t, conf=CONF):
    """so3_diffuser.py:201-209."""
    s = so3_sigma(t, conf)
    return np.sqrt(2 * (np.exp(conf["max_sigma"]) - np.exp(conf["min_sigma"])) * s / np.exp(s))


def igso3_series(omega, sigma, L=1000):
    """so3_diffuser.py:9-49 (expansion f) and :71-117 (d/domega numerator f').
    omega [...], sigma broadcastable; float64 torch tensors.  Returns (f, df)."""
    # dtype semantics follow the reference exactly: `ls` is int64, `ls + 1/2` is float32,
    # so with a float32 omega the sin/cos arguments and the quotient-rule bracket are
    # float32 while the Gaussian weights (float64 sigma) are float64.
    ls = torch.arange(L)
    om = omega[..., None]
    sg = sigma[..., None]
    lh = ls + 1 / 2
    w = (2 * ls + 1) * torch.exp(-ls * (ls + 1) * sg ** 2 / 2)
    hi = torch.sin(om * lh)
    dhi = lh * torch.cos(om * lh)
    lo = torch.sin(om / 2)
    dlo = 1 / 2 * torch.cos(om / 2)
    f = (w * hi / lo).sum(-1)
    df = (w * (lo * dhi - hi * dlo) / lo ** 2).sum(-1)
    return f, df


def so3_score_norms(conf=CONF, L=1000):
    """so3_diffuser.py:131-180: the cached table score_norms[num_sigma, num_omega] = f'/(f + 1e-4) on the
    discrete_omega = linspace(0, pi, num_omega + 1)[1:] grid (numpy float64, as precompute does)."""
    omega = np.linspace(0, np.pi, conf["num_omega"] + 1)[1:]
    sig = so3_discrete_sigma(conf)
    rows = []
    for i in range(0, len(sig), 20):                       # [20, num_omega, L] float64 temporaries
        f, df = igso3_series(torch.tensor(omega)[None, :], torch.tensor(sig[i:i + 20])[:, None], L)
        rows.append((df / (f + 1e-4)).numpy())
    return np.concatenate(rows, 0), omega


def so3_torch_score_cached(vec, t, score_norms, discrete_omega, conf=CONF, eps=1e-6):
    """so3_diffuser.py:293-299,305 (use_cached_score=True, config/icml_published.yaml): the score norm is
    score_norms[t_to_idx(t), bucketize(omega, discrete_omega[:-1])]; the lookup is a constant for autograd, so
    the gradient only flows through vec / (omega + eps)."""
    omega = torch.linalg.norm(vec, dim=-1) + eps
    rows = torch.tensor(np.asarray(score_norms)[so3_t_to_idx(t.detach().cpu().numpy(), conf)])
    rows = rows.reshape(rows.shape[0], -1)
    idx = torch.bucketize(omega.detach(), torch.tensor(np.asarray(discrete_omega)[:-1]))
    scal = torch.gather(rows, 1, idx.reshape(rows.shape[0], -1)).reshape(omega.shape)
    return scal[..., None] * vec / (omega[..., None] + eps)


def so3_torch_score(vec, t, conf=CONF, eps=1e-6):
    """so3_diffuser.py:274-305. vec [B,N,3]; t [B] -> float64 [B,N,3].  conf["score_norms"] /
    conf["discrete_omega"] select the cached branch."""
    if conf.get("score_norms") is not None:
        return so3_torch_score_cached(vec, t, conf["score_norms"], conf["discrete_omega"], conf, eps)
    omega = torch.linalg.norm(vec, dim=-1) + eps
    sig = so3_discrete_sigma(conf)[so3_t_to_idx(t.detach().cpu().numpy(), conf)]
    sig = torch.tensor(sig, dtype=torch.float64)[:, None]
    # float32 omega x float64 sigma -> float64 result (so3_diffuser.py:301-305)
    f, df = igso3_series(omega, sig)
    scal = df / (f + 1e-4)
    return scal[..., None] * vec / (omega[..., None] + eps)


def calc_rot_score(quat_t, quat_0, t, conf=CONF):
    """se3_diffuser.py:119-125: q_0t = inv(q_0) (x) q_t -> rotvec -> IGSO3 score.
    NB the caller (ipa_pytorch.py:650-654) passes rots_t = init (noised input) rotations
    and rots_0 = predicted rotations."""
    q0t = quat_multiply(invert_quat(quat_0), quat_t)
    return so3_torch_score(quat_to_rotvec(q0t), t, conf)


def r3_marginal_b_t(t, conf=CONF):
    """r3_diffuser.py:42-43."""
    return t * conf["min_b"] + 0.5 * t ** 2 * (conf["max_b"] - conf["min_b"])


def r3_b_t(t, conf=CONF):
    """r3_diffuser.py:26-29."""
    return conf["min_b"] + t * (conf["max_b"] - conf["min_b"])


def calc_trans_score(x_t, x_0, t, conf=CONF):
    """se3_diffuser.py:115-117 -> r3_diffuser.py:158-166 with scale=True, use_torch=True.
    x in Angstrom, t broadcastable ([B,1,1])."""
    sc = conf["coordinate_scaling"]
    beta = r3_marginal_b_t(t, conf)
    return -(x_t * sc - torch.exp(-0.5 * beta) * (x_0 * sc)) / (1 - torch.exp(-beta))


def r3_score_scaling(t, conf=CONF):
    """r3_diffuser.py:103-104."""
    return 1.0 / np.sqrt(1 - np.exp(-r3_marginal_b_t(np.asarray(t, dtype=np.float64), conf)))


# ----------------------------------------------------------------------------
# heads + backbone atoms
# ----------------------------------------------------------------------------
def torsion_head(P, s, pre="score_model.torsion_pred"):
    """ipa_pytorch.py:474-507."""
    h = F.relu(F.linear(s, P[f"{pre}.linear_1.weight"], P[f"{pre}.linear_1.bias"]))
    h = F.linear(h, P[f"{pre}.linear_2.weight"], P[f"{pre}.linear_2.bias"]) + s
    u = F.linear(h, P[f"{pre}.linear_final.weight"], P[f"{pre}.linear_final.bias"])
    return u / torch.sqrt(torch.clamp((u ** 2).sum(-1, keepdim=True), min=1e-8))


# ALA idealised backbone (data/residue_constants.py:127-133; frames :784-852)
_N = (-0.525, 1.363, 0.000)
_CA = (0.000, 0.000, 0.000)
_C = (1.526, -0.000, -0.000)
_CB = (-0.529, -0.774, -1.205)
_O = (0.627, 1.062, 0.000)


def _psi_frame():
    """Default frame 3 (psi group): built from ex = C - CA, ey = CA - N, origin C
    (residue_constants.py:819-824 via _make_rigid_transformation_4x4 :769-781)."""
    ex = np.array(_C) - np.array(_CA)
    ey = np.array(_CA) - np.array(_N)
    ex_n = ex / np.linalg.norm(ex)
    ey_n = ey - np.dot(ey, ex_n) * ex_n
    ey_n = ey_n / np.linalg.norm(ey_n)
    ez = np.cross(ex_n, ey_n)
    return np.stack([ex_n, ey_n, ez], 1), np.array(_C)


def backbone_atoms(quat, trans, psi):
    """all_atom.py:152-174 + feats.py:165-228 + all_atom.py:110-149 restricted to ALA.
    quat/trans in Angstrom frames; psi = (sin, cos).  -> atom37 [B,N,37,3], atom14 [B,N,14,3]."""
    R = quat_to_rot(quat)
    dt = quat.dtype
    Rd, td = _psi_frame()
    Rd = torch.tensor(Rd, dtype=torch.float32).to(dt)
    td = torch.tensor(td, dtype=torch.float32).to(dt)

    def place(local):
        p = torch.tensor(local, dtype=torch.float32).to(dt)
        return rot_apply(R, p.expand_as(trans)) + trans

    n, ca, c, cb = place(_N), place(_CA), place(_C), place(_CB)
    s, co = psi[..., 0], psi[..., 1]
    one, zero = torch.ones_like(s), torch.zeros_like(s)
    Rpsi = torch.stack([torch.stack([one, zero, zero], -1), torch.stack([zero, co, -s], -1),
                        torch.stack([zero, s, co], -1)], -2)
    # frame_to_bb = default_frame o psi-rotation ; to_global = bb o frame_to_bb
    R3 = Rd @ Rpsi
    Rg = R @ R3
    tg = rot_apply(R, td.expand_as(trans)) + trans
    o = rot_apply(Rg, torch.tensor(_O, dtype=torch.float32).to(dt).expand_as(trans)) + tg
    B, N = quat.shape[:2]
    atom14 = torch.zeros(B, N, 14, 3, dtype=dt)
    atom14[:, :, 0], atom14[:, :, 1], atom14[:, :, 2], atom14[:, :, 3], atom14[:, :, 4] = n, ca, c, o, cb
    atom37 = torch.zeros(B, N, 37, 3, dtype=dt)
    atom37[:, :, 0], atom37[:, :, 1], atom37[:, :, 2], atom37[:, :, 3], atom37[:, :, 4] = n, ca, c, cb, o
    return atom37, atom14


# ----------------------------------------------------------------------------
# full network  (score_network.py:170-215 + ipa_pytorch.py:611-672)
# ----------------------------------------------------------------------------
def score_network_forward(P, feats, conf=CONF, tfmr_mask_mode="additive", return_aux=False):
    dt = next(iter(P.values())).dtype
    mask = feats["res_mask"].to(dt)
    fixed = feats["fixed_mask"].to(dt)
    emask = mask[..., None] * mask[..., None, :]
    node, edge = embedder(P, feats["seq_idx"], feats["t"], fixed.float(), feats["sc_ca_t"], conf, dt)
    edge = edge * emask[..., None]
    node = node * mask[..., None]
    dmask = (1 - fixed) * mask
    frames = feats["rigids_t"].to(dt)
    quat0, trans0 = frames[..., :4], frames[..., 4:]
    sc = conf["coordinate_scaling"]
    quat, trans = quat0, trans0 * sc
    init_node = node * mask[..., None]
    node = init_node * mask[..., None]
    aux = {}
    pre0 = "score_model.trunk"
    if tfmr_mask_mode == "additive":
        key_add = 1 - mask
    else:
        key_add = torch.where(mask > 0, torch.zeros_like(mask), torch.full_like(mask, float("-inf")))
    for b in range(conf["num_blocks"]):
        ipa_out = ipa(P, f"{pre0}.ipa_{b}", node, edge, quat, trans, mask, conf) * mask[..., None]
        node = F.layer_norm(node + ipa_out, node.shape[-1:], P[f"{pre0}.ipa_ln_{b}.weight"], P[f"{pre0}.ipa_ln_{b}.bias"])
        skip = F.linear(init_node, P[f"{pre0}.skip_embed_{b}.weight"], P[f"{pre0}.skip_embed_{b}.bias"])
        x = torch.cat([node, skip], -1)
        for l in range(conf["tfmr_layers"]):
            x = tfmr_layer(P, f"{pre0}.seq_tfmr_{b}.layers.{l}", x, key_add, conf["tfmr_heads"])
        if tfmr_mask_mode != "additive":
            x = x * mask[..., None]
        node = node + F.linear(x, P[f"{pre0}.post_tfmr_{b}.weight"], P[f"{pre0}.post_tfmr_{b}.bias"])
        node = node_transition(P, f"{pre0}.node_transition_{b}", node) * mask[..., None]
        upd = F.linear(node * dmask[..., None], P[f"{pre0}.bb_update_{b}.linear.weight"], P[f"{pre0}.bb_update_{b}.linear.bias"])
        quat, trans = backbone_update(quat, trans, upd, dmask)
        if b < conf["num_blocks"] - 1:
            edge = edge_transition(P, f"{pre0}.edge_transition_{b}", node, edge) * emask[..., None]
        if return_aux:
            aux[f"node_{b}"] = node
            aux[f"quat_{b}"] = quat
            aux[f"trans_{b}"] = trans
            aux[f"edge_{b}"] = edge
    rot_score = calc_rot_score(quat0, quat, feats["t"], conf) * mask[..., None]
    trans_a = trans / sc
    trans_score = calc_trans_score(trans0, trans_a, feats["t"][:, None, None].to(dt) if feats["t"].dtype != torch.float64
                                   else feats["t"][:, None, None], conf) * mask[..., None]
    psi = torsion_head(P, node)
    gt_psi = feats["torsion_angles_sin_cos"][..., 2, :].to(psi.dtype)
    fm = fixed[..., None]
    psi = (1 - fm) * psi + fm * gt_psi
    rigids = torch.cat([quat, trans_a], -1)
    atom37, atom14 = backbone_atoms(quat, trans_a, psi)
    out = dict(psi=psi, rot_score=rot_score, trans_score=trans_score, rigids=rigids, atom37=atom37, atom14=atom14)
    if return_aux:
        out["aux"] = aux
    return out


# ----------------------------------------------------------------------------
# deterministic parameter / input synthesis shared by golden generation and tests
# ----------------------------------------------------------------------------
def param_shapes(conf=CONF):
    """state_dict names and shapes (verified against the reference in make_golden)."""
    cs, cz, H, C = conf["c_s"], conf["c_z"], conf["no_heads"], conf["c_hidden"]
    Pq, Pv, ck = conf["no_qk_points"], conf["no_v_points"], conf["c_skip"]
    shp = {}

    def lin(name, o, i, bias=True):
        shp[name + ".weight"] = (o, i)
        if bias:
            shp[name + ".bias"] = (o,)

    def ln(name, c):
        shp[name + ".weight"] = (c,)
        shp[name + ".bias"] = (c,)

    ne = "embedding_layer.node_embedder"
    ee = "embedding_layer.edge_embedder"
    nin = conf["index_embed_size"] * 2 + 1
    ein = (conf["index_embed_size"] + 1) * 2 + conf["index_embed_size"] + conf["num_bins"]
    lin(ne + ".0", cs, nin); lin(ne + ".2", cs, cs); lin(ne + ".4", cs, cs); ln(ne + ".5", cs)
    lin(ee + ".0", cz, ein); lin(ee + ".2", cz, cz); lin(ee + ".4", cz, cz); ln(ee + ".5", cz)
    d = cs + ck
    for b in range(conf["num_blocks"]):
        p = f"score_model.trunk.ipa_{b}"
        shp[p + ".head_weights"] = (H,)
        lin(p + ".linear_q", H * C, cs); lin(p + ".linear_kv", 2 * H * C, cs)
        lin(p + ".linear_q_points", H * Pq * 3, cs); lin(p + ".linear_kv_points", H * (Pq + Pv) * 3, cs)
        lin(p + ".linear_b", H, cz); lin(p + ".down_z", cz // 4, cz)
        lin(p + ".linear_out", cs, H * (cz // 4 + C + Pv * 4)); lin(p + ".linear_rbf", 1, 20)
        ln(f"score_model.trunk.ipa_ln_{b}", cs)
        lin(f"score_model.trunk.skip_embed_{b}", ck, cs)
        for l in range(conf["tfmr_layers"]):
            t = f"score_model.trunk.seq_tfmr_{b}.layers.{l}"
            shp[t + ".self_attn.in_proj_weight"] = (3 * d, d)
            shp[t + ".self_attn.in_proj_bias"] = (3 * d,)
            lin(t + ".self_attn.out_proj", d, d); lin(t + ".linear1", d, d); lin(t + ".linear2", d, d)
            ln(t + ".norm1", d); ln(t + ".norm2", d)
        lin(f"score_model.trunk.post_tfmr_{b}", cs, d)
        nt = f"score_model.trunk.node_transition_{b}"
        lin(nt + ".linear_1", cs, cs); lin(nt + ".linear_2", cs, cs); lin(nt + ".linear_3", cs, cs); ln(nt + ".ln", cs)
        lin(f"score_model.trunk.bb_update_{b}.linear", 6, cs)
        if b < conf["num_blocks"] - 1:
            e = f"score_model.trunk.edge_transition_{b}"
            hid = cs // 2 * 2 + cz
            lin(e + ".initial_embed", cs // 2, cs); lin(e + ".trunk.0", hid, hid); lin(e + ".trunk.2", hid, hid)
            lin(e + ".final_layer", cz, hid); ln(e + ".layer_norm", cz)
    tp = "score_model.torsion_pred"
    lin(tp + ".linear_1", cs, cs); lin(tp + ".linear_2", cs, cs); lin(tp + ".linear_3", cs, cs); lin(tp + ".linear_final", 2, cs)
    return shp


def synth_params(seed=0, conf=CONF, dtype=torch.float32):
    """Deterministic non-degenerate parameters (numpy MT19937: stable across versions).
    Weights ~ N(0, 1/fan_in) (no zero-initialised 'final' layers, so every branch of the
    path carries signal -- SURVEY.md 8d); biases ~ 0.1 N(0,1); LN gamma ~ 1 + 0.1 N."""
    rs = np.random.RandomState(seed)
    P = {}
    for name, shp in param_shapes(conf).items():
        x = rs.standard_normal(size=shp).astype(np.float32)
        if name.endswith("head_weights"):
            x = 0.5413 + 0.2 * x
        elif len(shp) == 2:
            x = x / math.sqrt(shp[1])
            if "bb_update" in name:
                x = x * 0.1
        else:
            is_ln = (".norm" in name or "ipa_ln" in name or ".ln." in name or "layer_norm" in name
                     or name.endswith("embedder.5.weight") or name.endswith("embedder.5.bias"))
            if is_ln and name.endswith("weight"):
                x = 1.0 + 0.1 * x
            else:
                x = 0.1 * x
                if "bb_update" in name:
                    x = x * 0.1
        P[name] = torch.tensor(x).to(dtype)
    return P


def synth_feats(B, N, seed=0, n_pad=0, n_fixed=0, t=None):
    """Synthetic ScoreNetwork inputs: random-walk CA trace (3.8 A steps),
    random unit quaternions, self-conditioning CA = noisy copy, 1-based seq_idx."""
    rs = np.random.RandomState(1000 + seed)
    steps = rs.standard_normal((B, N, 3))
    steps = 3.8 * steps / np.linalg.norm(steps, axis=-1, keepdims=True)
    ca = np.cumsum(steps, 1)
    ca = ca - ca.mean(1, keepdims=True)
    q = rs.standard_normal((B, N, 4))
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    res_mask = np.ones((B, N), np.float32)
    seq_idx = np.tile(np.arange(1, N + 1)[None], (B, 1))
    if n_pad:
        res_mask[:, N - n_pad:] = 0
        seq_idx[:, N - n_pad:] = 0
    fixed = np.zeros((B, N), np.float32)
    if n_fixed:
        fixed[:, :n_fixed] = 1
    tt = rs.uniform(0.05, 0.95, size=(B,)) if t is None else np.full((B,), t)
    sc = ca + rs.standard_normal((B, N, 3))
    tor = rs.standard_normal((B, N, 7, 2))
    tor = tor / np.linalg.norm(tor, axis=-1, keepdims=True)
    return dict(
        res_mask=torch.tensor(res_mask), fixed_mask=torch.tensor(fixed),
        seq_idx=torch.tensor(seq_idx, dtype=torch.int64), t=torch.tensor(tt, dtype=torch.float32),
        sc_ca_t=torch.tensor(sc, dtype=torch.float32),
        rigids_t=torch.tensor(np.concatenate([q, ca], -1), dtype=torch.float32),
        torsion_angles_sin_cos=torch.tensor(tor, dtype=torch.float32),
    )
