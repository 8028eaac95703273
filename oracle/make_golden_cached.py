"""TEST INFRASTRUCTURE -- golden vectors for so3.use_cached_score=True (config/icml_published.yaml) from the
UNMODIFIED reference; pins the oracle's cached branch (framediff_oracle.so3_torch_score_cached / so3_score_norms).

Run in the build container only (needs /root/reference):
    python oracle/make_golden_cached.py
Writes tests/golden/cached_score.npz.  Kept separate from make_golden.py so that regenerating it leaves the other
fixtures byte-identical.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader as rl  # noqa: E402
from oracle import framediff_oracle as fo  # noqa: E402
from oracle.make_golden import maxrel, GOLD, CACHE  # noqa: E402


def main():
    rl.install()
    from data import se3_diffuser  # reference modules
    from model import score_network
    from openfold.utils import rigid_utils as ru

    conf = rl.base_conf(CACHE, use_cached_score=True)
    diff = se3_diffuser.SE3Diffuser(conf.diffuser)
    so3 = diff._so3_diffuser
    assert so3.use_cached_score
    d = {}
    # the oracle's own table against the reference's precomputed one
    sn, om = fo.so3_score_norms()
    print("score_norms oracle-vs-ref", maxrel(sn, so3._score_norms), "omega", maxrel(om, so3.discrete_omega), flush=True)
    oconf = dict(fo.CONF, score_norms=so3._score_norms, discrete_omega=so3.discrete_omega)

    # torch_score lookups + gradient (fp32 vectors, as the network hands them over)
    rs = np.random.RandomState(15)
    ts = np.array([0.01, 0.05, 0.2, 0.5, 0.77, 1.0])
    vec = rs.standard_normal((6, 7, 3)).astype(np.float32)
    vec = vec / np.linalg.norm(vec, axis=-1, keepdims=True) * rs.uniform(1e-4, 3.1, size=(6, 7, 1)).astype(np.float32)
    vec[0, 0] = [1e-4, 0, 0]
    vec[1, 0] = [0, 3.14159, 0]
    vt = torch.tensor(vec, requires_grad=True)
    tt = torch.tensor(ts, dtype=torch.float32)
    sc = so3.torch_score(vt, tt)
    gw = torch.tensor(rs.standard_normal(sc.shape))
    (sc * gw).sum().backward()
    d["ts"], d["ts_vec"], d["ts_score"], d["ts_gw"], d["ts_grad"] = ts, vec, sc.detach().numpy(), gw.numpy(), vt.grad.numpy()
    vo = torch.tensor(vec, requires_grad=True)
    sco = fo.so3_torch_score(vo, tt, oconf)
    (sco * gw).sum().backward()
    print("torch_score(cached) oracle-vs-ref", maxrel(sco.detach(), sc.detach()), maxrel(vo.grad, vt.grad), flush=True)
    # NB: the reference's numpy score() / forward_marginal() cannot run with use_cached_score=True (torch.gather of a
    # [1, num_omega] row with a 1-D index raises), so the cached branch is pinned through torch_score,
    # calc_rot_score and the network only.
    # calc_rot_score through Rotation objects
    q_t = torch.tensor(rs.standard_normal((6, 7, 4)), dtype=torch.float32)
    q_t = q_t / q_t.norm(dim=-1, keepdim=True)
    q_0 = torch.tensor(rs.standard_normal((6, 7, 4)), dtype=torch.float32)
    q_0 = q_0 / q_0.norm(dim=-1, keepdim=True)
    rsr = diff.calc_rot_score(ru.Rotation(quats=q_t, normalize_quats=False), ru.Rotation(quats=q_0, normalize_quats=False), tt)
    d["crs_qt"], d["crs_q0"], d["crs_out"] = q_t.numpy(), q_0.numpy(), rsr.numpy()
    print("calc_rot_score(cached)", maxrel(fo.calc_rot_score(q_t, q_0, tt, oconf), rsr), flush=True)
    # ScoreNetwork forward + gradients with the cached rotation score (same case as fwd_n12_b2_pad_fixed)
    c = dict(B=2, N=12, seed=0, n_pad=2, n_fixed=3, blocks=4)
    torch.manual_seed(0)
    model = score_network.ScoreNetwork(conf.model, diff)
    P = fo.synth_params(seed=c["seed"], conf=dict(fo.CONF, num_blocks=c["blocks"]))
    model.load_state_dict(P, strict=True)
    feats = fo.synth_feats(c["B"], c["N"], seed=c["seed"], n_pad=c["n_pad"], n_fixed=c["n_fixed"])
    model.train()
    out = model({k: v.clone() for k, v in feats.items()})
    w = torch.tensor(np.random.RandomState(78).standard_normal(tuple(out["rot_score"].shape)))
    (out["rot_score"] * w).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    oo = fo.score_network_forward(Po, feats, dict(oconf, num_blocks=c["blocks"]), tfmr_mask_mode="additive")
    (oo["rot_score"] * w).sum().backward()
    gerrs = sorted(((maxrel(Po[n].grad if Po[n].grad is not None else torch.zeros_like(g), g), n, float(g.abs().max()))
                    for n, g in grads.items()), reverse=True)
    print("worst gradients (rel err, name, |g|max):", gerrs[:4], flush=True)
    gerr = gerrs[0][0]
    print("score_network(cached) rot_score", maxrel(oo["rot_score"].detach(), out["rot_score"].detach()), "grads", gerr, flush=True)
    d.update({"net_" + k: v for k, v in c.items()})
    d["net_rot_score"], d["net_w"] = out["rot_score"].detach().numpy(), w.numpy()
    for n, g in grads.items():
        if g.numel() <= 2048:
            d["grad/" + n] = g.numpy()
        else:
            d["gsig/" + n] = np.array([g.double().sum(), g.double().abs().sum(), g.double().norm()])
    np.savez_compressed(os.path.join(GOLD, "cached_score.npz"), **d)
    print("wrote", os.path.join(GOLD, "cached_score.npz"))


if __name__ == "__main__":
    main()
