"""fd_ipa_flash_fwd -- IPA attention of a trunk block in one launch (model/ipa_pytorch.py:380-457: logits, softmax, o, o_pt,
|o_pt|, o_pair) -- against (1) the launch sequence it replaces (q k^T GEMM -> fd_ipa_attn_fwd -> a v / a v_pts GEMMs ->
fd_ipa_opt_fwd, themselves oracle-tested) and (2) a float64 restatement of the reference's arithmetic on the same inputs.
Tolerances are relative to each output group's maximum and <= 10 x what the kernel achieves (recorded by parity_log)."""
import math

import pytest
import torch

import parity_log
from se3_diffusion_amd import ops

H, C, PQ, PV, ZB, CZ4 = 8, 256, 8, 12, 40, 32
LDP, LDF = 6816, 2688
F_PT, F_NORM, F_PAIR = 2048, 2336, 2432


def _inputs(dev, B, N, seed, spread=1.0, masked=True):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    R = B * N
    proj = rn(R, LDP)
    quat = rn(R, 4)
    quat = quat / quat.norm(dim=-1, keepdim=True)
    # translations in the diffuser's units (0.1 A): N(0, 1) per axis at t = 1 for every N; a folded 512-residue chain has a
    # radius of gyration of ~2.4 (1.4 per axis).  Drawn independently per residue here -- no chain locality, which is the worst
    # case for the tile-centred point term of the kernel (neighbours in sequence are not neighbours in space)
    trans = rn(R, 3, sc=spread)
    zb = rn(R * N, ZB)
    hw = rn(H)
    mask = ((torch.rand(R, generator=g) > 0.1).float() if masked else torch.ones(R)).to(dev)
    return proj, quat, trans, zb, hw, mask


def _points(L, proj, quat, trans, B, N):
    R = B * N
    qp = torch.empty(R, H, PQ * 3, device=proj.device); kp = torch.empty_like(qp)
    vp = torch.empty(R, H, PV * 3, device=proj.device)
    kpT = torch.empty(B, H, PQ * 3, N, device=proj.device)
    L.call("fd_ipa_points_fwd", proj, quat, trans, qp, kp, vp, kpT, N, R, H, C, PQ, PV)
    return qp, kp, vp, kpT


def _sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N):
    """network.ipa_fwd's launches (the path the flash kernel replaces)."""
    R = B * N
    dev = proj.device
    A = torch.empty(B, H, N, N, device=dev)
    L.gemm(proj, proj, A, N, N, C, (LDP, 1), (1, LDP), N, b_off=2048, batch=B * H, bdiv=H,
           a_bs=(N * LDP, C), b_bs=(N * LDP, 2 * C), c_bs=(H * N * N, N * N), alpha=math.sqrt(1.0 / (3 * C)))
    feats = torch.zeros(R, LDF, device=dev)
    L.call("fd_ipa_attn_fwd", A, zb, qp, kp, kpT, hw, mask, feats, B, N)
    L.gemm(A, proj, feats, N, C, N, (N, 1), (LDP, 1), LDF, b_off=2048 + C, batch=B * H, bdiv=H,
           a_bs=(H * N * N, N * N), b_bs=(N * LDP, 2 * C), c_bs=(N * LDF, C))
    optg = torch.empty(R, H, PV * 3, device=dev)
    L.gemm(A, vp, optg, N, PV * 3, N, (N, 1), (H * PV * 3, 1), H * PV * 3, batch=B * H, bdiv=H,
           a_bs=(H * N * N, N * N), b_bs=(N * H * PV * 3, PV * 3), c_bs=(N * H * PV * 3, PV * 3))
    L.call("fd_ipa_opt_fwd", optg, quat, trans, feats, R)
    return feats, A


def _quat_to_rot(q):
    a, b, c, d = q.unbind(-1)
    return torch.stack([a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c),
                        2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b),
                        2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d], -1).view(*q.shape[:-1], 3, 3)


def _float64(proj, quat, trans, zb, hw, mask, B, N):
    """ipa_pytorch.py:351-457 in float64 on the raw projections (points rotated here, not taken from the kernel)."""
    P = proj.double().cpu().view(B, N, LDP)
    Rm = _quat_to_rot(quat.double().cpu()).view(B, N, 3, 3)
    t = trans.double().cpu().view(B, N, 3)
    q = P[..., :2048].view(B, N, H, C)
    kv = P[..., 2048:6144].view(B, N, H, 2 * C)
    k, v = kv[..., :C], kv[..., C:]
    qpr = P[..., 6144:6336].view(B, N, 3, H * PQ).permute(0, 1, 3, 2)              # [B,N,64,3] raw
    kvpr = P[..., 6336:6816].view(B, N, 3, H * (PQ + PV)).permute(0, 1, 3, 2)      # [B,N,160,3]
    glob = lambda x: torch.einsum("bnac,bnpc->bnpa", Rm, x) + t[:, :, None, :]
    qpg = glob(qpr).view(B, N, H, PQ, 3)
    kvpg = glob(kvpr).view(B, N, H, PQ + PV, 3)
    kpg, vpg = kvpg[..., :PQ, :], kvpg[..., PQ:, :]
    zbd = zb.double().cpu().view(B, N, N, ZB)
    m = mask.double().cpu().view(B, N)
    gamma = torch.nn.functional.softplus(hw.double().cpu()) * math.sqrt(1.0 / (3.0 * (PQ * 9.0 / 2.0)))
    a = torch.einsum("bihc,bjhc->bhij", q, k) * math.sqrt(1.0 / (3 * C))
    a = a + math.sqrt(1.0 / 3) * zbd[..., :H].permute(0, 3, 1, 2)
    d2 = ((qpg[:, :, None] - kpg[:, None]) ** 2).sum((-1, -2))                     # [B,i,j,H]
    a = a - 0.5 * (d2 * gamma).permute(0, 3, 1, 2)
    a = a + 1e5 * (m[:, None, :, None] * m[:, None, None, :] - 1)
    a = torch.softmax(a, -1)
    o = torch.einsum("bhij,bjhc->bihc", a, v).reshape(B * N, H * C)
    og = torch.einsum("bhij,bjhpx->bihpx", a, vpg)
    ol = torch.einsum("bnca,bnhpc->bnhpa", Rm, og - t[:, :, None, None, :])        # R^T (o - t)
    nrm = torch.sqrt((ol ** 2).sum(-1) + 1e-8)
    opair = torch.einsum("bhij,bijc->bihc", a, zbd[..., H:]).reshape(B * N, H * CZ4)
    ol = ol.reshape(B * N, H * PV, 3)
    feats = torch.cat([o, ol[..., 0], ol[..., 1], ol[..., 2], nrm.reshape(B * N, H * PV), opair], -1)
    return feats, a


GROUPS = (("o", 0, F_PT), ("o_pt", F_PT, F_NORM), ("norm", F_NORM, F_PAIR), ("o_pair", F_PAIR, LDF))


def _run(dev, B, N, seed, hpb=0, spread=1.0, want_A=True, tol=2e-5, tol_seq=2e-5, log=None, splits=1):
    L = ops.lib()
    proj, quat, trans, zb, hw, mask = _inputs(dev, B, N, seed, spread)
    qp, kp, vp, kpT = _points(L, proj, quat, trans, B, N)
    f_seq, A_seq = _sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N)
    feats = torch.full((B * N, LDF), float("nan"), device=dev)          # (every column must be written)
    A = torch.full((B, H, N, N), float("nan"), device=dev) if want_A else None
    if splits > 1:
        part = torch.full((splits * B * N * H * 328,), float("nan"), device=dev)
        L.call("fd_ipa_flash_fwd_split", proj, zb, qp, kp, vp, hw, mask, quat, trans, feats, None, B, N, hpb, splits, part)
        want_A = False
    else:
        L.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, hw, mask, quat, trans, feats, A, B, N, hpb)
    assert bool(torch.isfinite(feats).all())
    f64, a64 = _float64(proj, quat, trans, zb, hw, mask, B, N)
    rows = (mask.cpu() > 0)                                               # rows of masked residues: see test_ipa_attn.py
    out = {}
    for name, lo, hi in GROUPS:
        ref = f64[rows][:, lo:hi]
        sc = float(ref.abs().max())
        e64 = float((feats.cpu().double()[rows][:, lo:hi] - ref).abs().max()) / sc
        eseq = float((feats.cpu()[rows][:, lo:hi] - f_seq.cpu()[rows][:, lo:hi]).abs().max()) / sc
        s64 = float((f_seq.cpu().double()[rows][:, lo:hi] - ref).abs().max()) / sc      # what the sequence itself achieves
        out[name] = (e64, eseq, s64)
        if log is not None:
            parity_log.out(f"flash.{name}", e64)
        assert e64 < tol, (name, e64, s64)
        assert eseq < tol_seq, (name, eseq)
    if want_A:
        assert bool(torch.isfinite(A).all())
        rr = rows.view(B, 1, N, 1)
        ea = float(((A.cpu().double() - a64) * rr).abs().max())
        es = float(((A.cpu() - A_seq.cpu()) * rr).abs().max())
        if log is not None:
            parity_log.out("flash.A", ea)
        assert ea < tol and es < tol_seq, (ea, es)
    return out


def test_ipa_flash_fwd_emu(use_emu):
    _run("cpu", 1, 12, 0, hpb=8)          # one ragged tile
    _run("cpu", 2, 37, 1, hpb=4)          # three tiles, ragged rows and keys, two head groups
    _run("cpu", 1, 33, 2, hpb=2)
    _run("cpu", 1, 20, 3, hpb=0, want_A=False)
    _run("cpu", 1, 37, 4, hpb=4, splits=3)          # key split: 3 blocks per query tile + merge launch
    _run("cpu", 1, 33, 5, hpb=0, splits=8)          # (more splits than key tiles: clamped)


@pytest.mark.gpu
def test_ipa_flash_fwd_gpu(hip_lib):
    with parity_log.case("ipa_flash_fwd") as log:
        for (B, N, seed, hpb, spread) in ((2, 128, 0, 8, 1.0), (1, 128, 1, 2, 1.0), (1, 128, 2, 4, 1.0), (3, 100, 3, 8, 1.0),
                                          (1, 256, 4, 0, 1.2), (1, 257, 5, 8, 1.2), (2, 400, 6, 0, 1.5), (1, 512, 7, 8, 1.5),
                                          (1, 600, 8, 8, 1.5)):
            _run("cuda", B, N, seed, hpb=hpb, spread=spread, log=log)
        for (B, N, seed, hpb, ks) in ((1, 128, 10, 4, 4), (1, 128, 11, 4, 2), (1, 256, 12, 4, 4), (1, 256, 13, 4, 8), (1, 300, 14, 2, 4),
                                      (1, 512, 15, 4, 8)):
            _run("cuda", B, N, seed, hpb=hpb, spread=1.2, log=log, splits=ks)


# ------------------------------------------------------------------------------------------------------------------- backward
def _bwd_sequence(L, proj, quat, zb, hw, qp, kp, vp, kpT, A, feats, dfeats, B, N):
    """network.ipa_bwd's launches between dfeats and (dL, dzb, dqp, dkp, dhead_w): dA = dO V^T, fd_ipa_opt_bwd,
    dA += dOpt vpts^T, fd_ipa_attn_bwd."""
    R = B * N
    dev = proj.device
    dA = torch.empty(B, H, N, N, device=dev)
    L.gemm(dfeats, proj, dA, N, N, C, (LDF, 1), (1, LDP), N, b_off=2048 + C, batch=B * H, bdiv=H,
           a_bs=(N * LDF, C), b_bs=(N * LDP, 2 * C), c_bs=(H * N * N, N * N))
    doptg = torch.empty(R, H, PV * 3, device=dev)
    dframe = torch.zeros(R, 12, device=dev)
    L.call("fd_ipa_opt_bwd", dfeats, feats, quat, doptg, dframe, R)
    L.gemm(doptg, vp, dA, N, N, PV * 3, (H * PV * 3, 1), (1, H * PV * 3), N, batch=B * H, bdiv=H,
           a_bs=(N * H * PV * 3, PV * 3), b_bs=(N * H * PV * 3, PV * 3), c_bs=(H * N * N, N * N), beta=True)
    dzb = torch.empty(R * N, ZB, device=dev)
    dqp = torch.empty(R, H, PQ * 3, device=dev); dkp = torch.empty(R, H, PQ * 3, device=dev)
    dhw = torch.zeros(H, device=dev); part = torch.empty(R, H, device=dev)
    L.call("fd_ipa_attn_bwd", A, dA, zb, dfeats, qp, kp, kpT, hw, dzb, dqp, dkp, dhw, part, B, N)
    return dict(dL=dA, dzb=dzb, dqp=dqp, dkp=dkp, dhw=dhw, dframe=dframe)


def _bwd_flash(L, proj, quat, trans, zb, hw, qp, kp, vp, A, feats, dfeats, B, N, poison=True):
    R = B * N
    dev = proj.device
    doptg = torch.empty(R, H, PV * 3, device=dev)
    dframe = torch.zeros(R, 12, device=dev)
    ptdot = torch.empty(R, H, device=dev)
    L.call("fd_ipa_opt_bwd_dot", dfeats, feats, quat, trans, doptg, dframe, ptdot, R)
    # (poison: every element of the outputs must be written; the microbenchmark allocates like the launch sequence instead)
    new = (lambda *sh: torch.full(sh, float("nan"), device=dev)) if poison else (lambda *sh: torch.empty(sh, device=dev))
    dL = new(B, H, N, N)
    dzb = new(R * N, ZB)
    dqp = new(R, H, PQ * 3); dkp = new(R, H, PQ * 3)
    dhw = torch.zeros(H, device=dev); part = torch.empty(R, H, device=dev)
    L.call("fd_ipa_flash_bwd", proj, A, zb, dfeats, feats, doptg, ptdot, qp, kp, vp, hw, trans, dL, dzb, dqp, dkp, dhw, part, B, N)
    return dict(dL=dL, dzb=dzb, dqp=dqp, dkp=dkp, dhw=dhw, dframe=dframe)


def _run_bwd(dev, B, N, seed, spread=1.0, tol=2e-5, log=False):
    L = ops.lib()
    proj, quat, trans, zb, hw, mask = _inputs(dev, B, N, seed, spread)
    qp, kp, vp, kpT = _points(L, proj, quat, trans, B, N)
    feats, A = _sequence(L, proj, quat, trans, zb, hw, mask, qp, kp, vp, kpT, B, N)
    g = torch.Generator().manual_seed(seed + 100)
    dfeats = torch.randn(B * N, LDF, generator=g).to(dev)
    ref = _bwd_sequence(L, proj, quat, zb, hw, qp, kp, vp, kpT, A, feats, dfeats, B, N)
    out = _bwd_flash(L, proj, quat, trans, zb, hw, qp, kp, vp, A, feats, dfeats, B, N)
    errs = {}
    for k in ref:
        assert bool(torch.isfinite(out[k]).all()), k
        sc = float(ref[k].abs().max()) + 1e-30
        errs[k] = float((out[k] - ref[k]).abs().max()) / sc
        if log:
            parity_log.out(f"flash_bwd.{k}", errs[k])
    for k, e in errs.items():
        assert e < tol, (k, e, errs)
    return errs


def test_ipa_flash_bwd_emu(use_emu):
    _run_bwd("cpu", 1, 12, 0)
    _run_bwd("cpu", 2, 37, 1)
    _run_bwd("cpu", 1, 18, 2)            # N % 4 != 0: scalar rows of A / dL


@pytest.mark.gpu
def test_ipa_flash_bwd_gpu(hip_lib):
    with parity_log.case("ipa_flash_bwd"):
        for (B, N, seed, spread) in ((2, 128, 0, 1.0), (3, 100, 1, 1.0), (1, 256, 2, 1.2), (1, 257, 3, 1.2), (2, 400, 4, 1.5),
                                     (1, 512, 5, 1.5)):
            _run_bwd("cuda", B, N, seed, spread=spread, log=True)


# --------------------------------------------------------------------------------------------------------- backward, key side
def _run_bwd_keys(dev, B, N, seed, hpb=0, tol=5e-6, log=False):
    """fd_ipa_flash_bwd_keys (dV, dK, dv_pts, dk_pts of a block in one launch) against float64 contractions of the same A, dL --
    autograd of model/ipa_pytorch.py:380-457 with respect to keys / values: dV = A^T dO, dvp = A^T dO_pt, dK = sqrt(1/3C) dL^T Q,
    dkp = gamma sum_i dL_ij (qp_i - kp_j).  Columns of dproj the kernel does not own must stay untouched."""
    L = ops.lib()
    g = torch.Generator().manual_seed(seed)
    rn = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    R = B * N
    A = torch.softmax(rn(B, H, N, N), -1).contiguous()
    dL = (rn(B, H, N, N) * 0.1).contiguous()
    proj, dfeats, doptg = rn(R, LDP), rn(R, LDF), rn(R, H, PV * 3)
    qp, kp, hw = rn(R, H, PQ * 3), rn(R, H, PQ * 3), rn(H)
    dproj = torch.full((R, LDP), 7.0, device=dev)
    dvp = torch.full((R, H, PV * 3), float("nan"), device=dev)
    dkp = torch.full((R, H, PQ * 3), float("nan"), device=dev)
    L.call("fd_ipa_flash_bwd_keys", A, dL, proj, dfeats, doptg, qp, kp, hw, dproj, dvp, dkp, B, N, hpb)
    d = lambda t: t.double().cpu()
    A6, dL6 = d(A), d(dL)
    dO = d(dfeats)[:, :H * C].view(B, N, H, C)
    Q = d(proj)[:, :H * C].view(B, N, H, C)
    ref_dV = torch.einsum("bhij,bihc->bjhc", A6, dO)
    ref_dK = math.sqrt(1.0 / (3 * C)) * torch.einsum("bhij,bihc->bjhc", dL6, Q)
    ref_dvp = torch.einsum("bhij,bihc->bjhc", A6, d(doptg).view(B, N, H, PV * 3))
    gamma = torch.nn.functional.softplus(d(hw)) * math.sqrt(1.0 / (3 * (PQ * 9.0 / 2)))
    qp6, kp6 = d(qp).view(B, N, H, PQ * 3), d(kp).view(B, N, H, PQ * 3)
    ref_dkp = gamma[None, None, :, None] * (torch.einsum("bhij,bihc->bjhc", dL6, qp6) - kp6 * dL6.sum(2).permute(0, 2, 1)[..., None])
    kv = d(dproj)[:, 2048:2048 + 2 * H * C].view(B, N, H, 2, C)
    errs = {}
    for name, got, ref in (("dK", kv[..., 0, :], ref_dK), ("dV", kv[..., 1, :], ref_dV), ("dvp", d(dvp).view(B, N, H, -1), ref_dvp),
                           ("dkp", d(dkp).view(B, N, H, -1), ref_dkp)):
        assert bool(torch.isfinite(got).all()), name
        errs[name] = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        if log:
            parity_log.out(f"flash_bwd_keys.{name}", errs[name])
    other = torch.cat([d(dproj)[:, :2048], d(dproj)[:, 2048 + 2 * H * C:]], 1)
    assert bool((other == 7.0).all()), "columns outside dK / dV were written"
    for k, e in errs.items():
        assert e < tol, (k, e, errs)
    return errs


def test_ipa_flash_bwd_keys_emu(use_emu):
    _run_bwd_keys("cpu", 1, 16, 0, hpb=2)
    _run_bwd_keys("cpu", 1, 16, 3, hpb=1)        # (1: four key tiles of one head per block, operands through LDS)
    _run_bwd_keys("cpu", 2, 70, 4, hpb=1)        # 5 key tiles: a block with three idle waves, ragged rows
    _run_bwd_keys("cpu", 1, 37, 5, hpb=0)
    _run_bwd_keys("cpu", 2, 37, 1, hpb=4)        # ragged: rows and keys past N
    _run_bwd_keys("cpu", 1, 18, 2, hpb=8)


@pytest.mark.gpu
def test_ipa_flash_bwd_keys_gpu(hip_lib):
    with parity_log.case("ipa_flash_bwd_keys"):
        for (B, N, seed, hpb) in ((2, 128, 0, 0), (30, 128, 1, 0), (3, 100, 2, 2), (1, 257, 3, 4), (2, 400, 4, 8), (1, 512, 5, 0),
                                  (12, 200, 6, 1), (3, 101, 7, 1)):
            _run_bwd_keys("cuda", B, N, seed, hpb=hpb, log=True)


# ------------------------------------------------------------------------------------------------- inside the network forward
def _forward_paths(dev, B, N, blocks=2):
    """the eval-mode ScoreNetwork forward with IPA attention as (a) the launch sequence, (b) the one-launch kernel, (c) the
    key-split kernel + merge launch: same outputs (model/ipa_pytorch.py:380-457 inside IpaScore.forward)"""
    from se3_diffusion_amd import options, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    m = ScoreNetwork(ts.base_model_conf(blocks), diffuser=None).to(dev).eval()
    ts.perturb_final_layers(m, seed=0)
    batch = ts.synthetic_batch(B, N, dev, seed=3)
    outs = []
    with torch.no_grad():
        for kw in (dict(flash_ipa=False), dict(flash_ipa_min_tiles=0), dict(flash_ipa_min_tiles=1 << 30, flash_ipa_split_min_n=16)):
            with options.override(**kw):
                outs.append({k: v.clone() for k, v in m(batch).items() if torch.is_tensor(v)})
    for o in outs[1:]:
        # (psi is a normalised 2-vector: a residue whose unnormalised torsion output is small amplifies the last-bit differences
        #  between the summation orders of the paths -- profiles/r04_parity_errors.md has one such case at 1.4e-4)
        for k, tol in (("rot_score", 2e-5), ("trans_score", 2e-5), ("rigids", 2e-5), ("atom37", 2e-5), ("psi", 2e-3)):
            sc = float(outs[0][k].abs().max()) + 1e-12
            err = float((o[k] - outs[0][k]).abs().max()) / sc
            assert err < tol, (k, err)


def test_forward_paths_emu(use_emu):
    _forward_paths("cpu", 1, 24, blocks=1)


@pytest.mark.gpu
def test_forward_paths_gpu(hip_lib):
    _forward_paths("cuda", 1, 400, blocks=2)
    _forward_paths("cuda", 2, 128, blocks=2)


# ------------------------------------------------------------------------------------- size-independent properties, full sizes
@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(30, 128), (8, 512)])
def test_ipa_flash_properties_full_size(hip_lib, B, N):
    """At the benchmark sizes (no float64 restatement of a [B, 8, N, N] attention on the host):
    * the probabilities the training forward writes are a distribution over the keys of every (row, head);
    * attention is a SET function of the keys: permuting the residues of every backbone as keys only (K, V, key points, value
      points, the key axis of zb, the key mask) leaves o, o_pt, |o_pt|, o_pair where they were (the kernel walks the keys in
      tiles with a running maximum: any order must give the same sums up to fp32 reassociation)."""
    L = ops.lib()
    dev = "cuda"
    proj, quat, trans, zb, hw, mask = _inputs(dev, B, N, 21, 1.2)
    qp, kp, vp, _ = _points(L, proj, quat, trans, B, N)
    R = B * N
    feats = torch.empty(R, LDF, device=dev)
    A = torch.empty(B, H, N, N, device=dev)
    L.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, hw, mask, quat, trans, feats, A, B, N, 0)
    rows = (mask > 0).view(B, 1, N)
    assert float(((A.sum(-1) - 1.0).abs() * rows).max()) < 1e-5
    assert float(A.min()) >= 0.0
    # keys permuted: the K | V columns of proj, kp, vp, mask and zb's key axis follow the permutation, the query side stays
    g = torch.Generator().manual_seed(5)
    perm = torch.randperm(N, generator=g).to(dev)
    pr = proj.view(B, N, LDP)
    proj2 = pr.clone()
    proj2[:, :, 2048:6144] = pr[:, perm, 2048:6144]
    proj2 = proj2.view(R, LDP).contiguous()
    kp2 = kp.view(B, N, H, PQ * 3)[:, perm].reshape(R, H, PQ * 3).contiguous()
    vp2 = vp.view(B, N, H, PV * 3)[:, perm].reshape(R, H, PV * 3).contiguous()
    zb2 = zb.view(B, N, N, ZB)[:, :, perm].reshape(R * N, ZB).contiguous()
    # the mask enters as m_i m_j: the row factor must keep the QUERY order, so the comparison uses an all-ones mask
    ones = torch.ones_like(mask)
    f1 = torch.empty(R, LDF, device=dev); f2 = torch.empty(R, LDF, device=dev)
    L.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, hw, ones, quat, trans, f1, None, B, N, 0)
    L.call("fd_ipa_flash_fwd", proj2, zb2, qp, kp2, vp2, hw, ones, quat, trans, f2, None, B, N, 0)
    for name, lo, hi in GROUPS:
        sc = float(f1[:, lo:hi].abs().max())
        assert float((f1[:, lo:hi] - f2[:, lo:hi]).abs().max()) / sc < 1e-5, name
