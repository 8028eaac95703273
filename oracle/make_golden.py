"""TEST INFRASTRUCTURE -- generate tests/golden/*.npz from the UNMODIFIED reference and
pin oracle/framediff_oracle.py against it.

Run in the build container only (needs /root/reference):
    python oracle/make_golden.py
The reference ships no tests/golden vectors, so these fixtures --
outputs of the reference itself (torch 2.10 / numpy 2.2 / scipy 1.15 here; the reference
pins torch 1.13.1 / numpy 1.22.4 / scipy 1.7.3) -- are the definition of truth.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader as rl  # noqa: E402
from oracle import framediff_oracle as fo  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CACHE = os.environ.get("FD_IGSO3_CACHE", "/tmp/fd_igso3_cache")


def maxrel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def quat_sign_align(a, b):
    s = torch.sign((a[..., :4] * b[..., :4]).sum(-1, keepdim=True))
    return torch.cat([a[..., :4] * s, a[..., 4:]], -1)


def main():
    rl.install()
    from data import se3_diffuser, utils as du, all_atom  # reference modules
    from model import score_network
    from openfold.utils import rigid_utils as ru

    os.makedirs(GOLD, exist_ok=True)
    conf = rl.base_conf(CACHE)
    diff = se3_diffuser.SE3Diffuser(conf.diffuser)
    report = {}

    # ---------------- ScoreNetwork forward (+ gradients) ----------------
    cases = [
        dict(name="fwd_n12_b2_pad_fixed", B=2, N=12, seed=0, n_pad=2, n_fixed=3, blocks=4),
        dict(name="fwd_n24_b1", B=1, N=24, seed=1, n_pad=0, n_fixed=0, blocks=4),
        dict(name="fwd_n64_b1_1block", B=1, N=64, seed=2, n_pad=0, n_fixed=0, blocks=1),
    ]
    for c in cases:
        mconf = rl.base_conf(CACHE, num_blocks=c["blocks"]).model
        oconf = dict(fo.CONF, num_blocks=c["blocks"])
        torch.manual_seed(0)
        model = score_network.ScoreNetwork(mconf, diff)
        P = fo.synth_params(seed=c["seed"], conf=oconf)
        model.load_state_dict(P, strict=True)       # pins names + shapes
        feats = fo.synth_feats(c["B"], c["N"], seed=c["seed"], n_pad=c["n_pad"], n_fixed=c["n_fixed"])
        model.train()
        t0 = time.time()
        out = model({k: v.clone() for k, v in feats.items()})
        t_ref = time.time() - t0
        # loss = fixed random projection of the differentiable outputs
        rs = np.random.RandomState(77 + c["seed"])
        wts = {k: torch.tensor(rs.standard_normal(tuple(out[k].shape))).to(out[k].dtype)
               for k in ["rot_score", "trans_score", "rigids", "atom37", "psi"]}
        loss = sum((out[k] * wts[k]).sum() for k in wts)
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        # oracle on the same inputs
        Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        t0 = time.time()
        oo = fo.score_network_forward(Po, feats, oconf, tfmr_mask_mode="additive")
        t_or = time.time() - t0
        lo = sum((oo[k] * wts[k]).sum() for k in wts)
        lo.backward()
        rep = {}
        for k in ["psi", "rot_score", "trans_score", "atom37", "atom14"]:
            rep[k] = maxrel(oo[k].detach(), out[k].detach())
        rep["rigids"] = maxrel(quat_sign_align(oo["rigids"].detach(), out["rigids"].detach()), out["rigids"].detach())
        gerr = {}
        for n, g in grads.items():
            go = Po[n].grad
            gerr[n] = maxrel(go if go is not None else torch.zeros_like(g), g)
        worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:5]
        rep["grad_worst"] = worst
        rep["loss"] = (float(loss), float(lo))
        rep["t_ref_s"], rep["t_oracle_s"] = t_ref, t_or
        report[c["name"]] = rep
        print(c["name"], {k: (v if not isinstance(v, float) else f"{v:.2e}") for k, v in rep.items()}, flush=True)
        # eval/no-grad fast path of nn.TransformerEncoder (boolean key padding)
        model.eval()
        with torch.no_grad():
            out_eval = model({k: v.clone() for k, v in feats.items()})
        with torch.no_grad():
            oo_eval = fo.score_network_forward(P, feats, oconf, tfmr_mask_mode="bool")
        rep["eval_rigids"] = maxrel(quat_sign_align(oo_eval["rigids"], out_eval["rigids"]), out_eval["rigids"])
        rep["eval_rot_score"] = maxrel(oo_eval["rot_score"], out_eval["rot_score"])
        print("   eval-mode:", rep["eval_rigids"], rep["eval_rot_score"], flush=True)
        save = dict(B=c["B"], N=c["N"], seed=c["seed"], n_pad=c["n_pad"], n_fixed=c["n_fixed"], blocks=c["blocks"])
        for k, v in out.items():
            save["out_" + k] = v.detach().numpy()
        for k, v in out_eval.items():
            save["eval_" + k] = v.detach().numpy()
        for k, v in wts.items():
            save["w_" + k] = v.numpy()
        save["loss"] = float(loss)
        # gradients: small tensors in full, the rest as (sum, abs-sum, l2) signatures
        for n, g in grads.items():
            if g.numel() <= 2048:
                save["grad/" + n] = g.numpy()
            else:
                save["gsig/" + n] = np.array([g.double().sum(), g.double().abs().sum(), g.double().norm()])
        np.savez_compressed(os.path.join(GOLD, c["name"] + ".npz"), **save)

    # ---------------- SE(3) diffuser ----------------
    so3 = diff._so3_diffuser
    r3 = diff._r3_diffuser
    ts = np.array([0.01, 0.05, 0.2, 0.5, 0.77, 1.0])
    d = dict(ts=ts)
    d["sigma"] = np.array([so3.sigma(t) for t in ts])
    d["t_to_idx"] = np.array([so3.t_to_idx(t) for t in ts])
    d["g_rot"] = np.array([so3.diffusion_coef(t) for t in ts])
    d["rot_score_scaling"] = np.array([so3.score_scaling(t) for t in ts])
    d["trans_score_scaling"] = np.array([r3.score_scaling(t) for t in ts])
    d["b_t"] = np.array([r3.b_t(t) for t in ts])
    d["marginal_b_t"] = np.array([r3.marginal_b_t(t) for t in ts])
    d["discrete_sigma_sample"] = so3.discrete_sigma[[0, 1, 499, 998, 999]]
    d["score_scaling_table_sample"] = so3._score_scaling[[0, 9, 499, 999]]
    rows = [0, 9, 100, 499, 999]
    cols = [0, 1, 99, 499, 998, 999]
    d["tab_rows"], d["tab_cols"] = np.array(rows), np.array(cols)
    d["cdf_sample"] = so3._cdf[np.ix_(rows, cols)]
    d["pdf_sample"] = so3._pdf[np.ix_(rows, cols)]
    d["score_norms_sample"] = so3._score_norms[np.ix_(rows, cols)]
    d["cdf_row_499"] = so3._cdf[499]
    # torch_score (differentiable rot score) on a grid incl. tiny angles
    rs = np.random.RandomState(5)
    vec = rs.standard_normal((6, 7, 3)).astype(np.float32)
    vec = vec / np.linalg.norm(vec, axis=-1, keepdims=True) * rs.uniform(1e-4, 3.1, size=(6, 7, 1)).astype(np.float32)
    vec[0, 0] = [1e-4, 0, 0]
    vt = torch.tensor(vec, requires_grad=True)
    tt = torch.tensor(ts, dtype=torch.float32)
    sc = so3.torch_score(vt, tt)
    gw = torch.tensor(rs.standard_normal(sc.shape))
    (sc * gw).sum().backward()
    d["ts_vec"], d["ts_score"], d["ts_gw"], d["ts_grad"] = vec, sc.detach().numpy(), gw.numpy(), vt.grad.numpy()
    vo = torch.tensor(vec, requires_grad=True)
    sco = fo.so3_torch_score(vo, tt)
    (sco * gw).sum().backward()
    report["torch_score"] = (maxrel(sco.detach(), sc.detach()), maxrel(vo.grad, vt.grad))
    print("torch_score oracle-vs-ref", report["torch_score"], flush=True)
    # calc_rot_score / calc_trans_score through Rotation objects
    q_t = torch.tensor(rs.standard_normal((6, 7, 4)), dtype=torch.float32)
    q_t = q_t / q_t.norm(dim=-1, keepdim=True)
    q_0 = torch.tensor(rs.standard_normal((6, 7, 4)), dtype=torch.float32)
    q_0 = q_0 / q_0.norm(dim=-1, keepdim=True)
    rsr = diff.calc_rot_score(ru.Rotation(quats=q_t, normalize_quats=False), ru.Rotation(quats=q_0, normalize_quats=False), tt)
    d["crs_qt"], d["crs_q0"], d["crs_out"] = q_t.numpy(), q_0.numpy(), rsr.numpy()
    report["calc_rot_score"] = maxrel(fo.calc_rot_score(q_t, q_0, tt), rsr)
    x_t = torch.tensor(rs.standard_normal((6, 7, 3)) * 10, dtype=torch.float32)
    x_0 = torch.tensor(rs.standard_normal((6, 7, 3)) * 10, dtype=torch.float32)
    tsr = diff.calc_trans_score(x_t, x_0, tt[:, None, None], use_torch=True)
    d["cts_xt"], d["cts_x0"], d["cts_out"] = x_t.numpy(), x_0.numpy(), tsr.numpy()
    report["calc_trans_score"] = maxrel(fo.calc_trans_score(x_t, x_0, tt[:, None, None]), tsr)
    print("calc_rot/trans_score", report["calc_rot_score"], report["calc_trans_score"], flush=True)

    # reverse step with the reference's own numpy RNG stream (rot noise first, then trans)
    Bn, Nn = 2, 9
    rig = fo.synth_feats(Bn, Nn, seed=9)["rigids_t"]
    rot_score = rs.standard_normal((Bn, Nn, 3)) * 0.5
    trans_score = rs.standard_normal((Bn, Nn, 3)) * 0.5
    dmask = np.ones((Bn, Nn))
    dmask[:, :2] = 0
    for tag, t_, ns_ in [("a", 0.6, 1.0), ("b", 0.03, 0.1)]:
        np.random.seed(123)
        out_r = diff.reverse(rigid_t=ru.Rigid.from_tensor_7(rig.clone()), rot_score=rot_score, trans_score=trans_score,
                             t=t_, dt=1 / 100, diffuse_mask=dmask, center=True, noise_scale=ns_)
        np.random.seed(123)
        z_rot = np.random.normal(size=rot_score.shape)
        z_trans = np.random.normal(size=trans_score.shape)
        d[f"rev_{tag}_t"], d[f"rev_{tag}_ns"] = t_, ns_
        d[f"rev_{tag}_zrot"], d[f"rev_{tag}_ztrans"] = z_rot, z_trans
        d[f"rev_{tag}_out_rotmats"] = out_r.get_rots().get_rot_mats().numpy()
        d[f"rev_{tag}_out_trans"] = out_r.get_trans().numpy()
        d[f"rev_{tag}_out_t7"] = out_r.to_tensor_7().numpy()
    d["rev_rigids"], d["rev_rot_score"], d["rev_trans_score"], d["rev_dmask"] = rig.numpy(), rot_score, trans_score, dmask
    # sample_ref and forward_marginal with the reference RNG stream
    np.random.seed(321)
    sr = diff.sample_ref(n_samples=11, as_tensor_7=True)["rigids_t"]
    np.random.seed(321)
    d["sr_randn"] = np.random.randn(11, 3)
    d["sr_rand"] = np.random.rand(11)
    d["sr_normal"] = np.random.normal(size=(11, 3))
    d["sr_out_t7"] = sr.numpy()
    rig0 = ru.Rigid.from_tensor_7(fo.synth_feats(1, 10, seed=4)["rigids_t"][0])
    np.random.seed(55)
    fm = diff.forward_marginal(rig0, 0.37, diffuse_mask=None, as_tensor_7=True)
    np.random.seed(55)
    d["fm_randn"] = np.random.randn(10, 3)
    d["fm_rand"] = np.random.rand(10)
    d["fm_normal"] = np.random.normal(size=(10, 3))
    d["fm_rigids0"] = rig0.to_tensor_7().numpy()
    d["fm_t"] = 0.37
    d["fm_rigids_t"] = fm["rigids_t"].numpy()
    d["fm_trans_score"], d["fm_rot_score"] = fm["trans_score"], fm["rot_score"]
    d["fm_trans_score_scaling"], d["fm_rot_score_scaling"] = fm["trans_score_scaling"], fm["rot_score_scaling"]
    # compute_backbone
    psi = torch.tensor(rs.standard_normal((1, 10, 2)), dtype=torch.float32)
    psi = psi / psi.norm(dim=-1, keepdim=True)
    a37, _, _, a14 = all_atom.compute_backbone(ru.Rigid.from_tensor_7(d["fm_rigids0"][None] * 1.0 if False else torch.tensor(d["fm_rigids0"])[None]), psi)
    d["bb_psi"], d["bb_atom37"], d["bb_atom14"] = psi.numpy(), a37.numpy(), a14.numpy()
    o37, o14 = fo.backbone_atoms(torch.tensor(d["fm_rigids0"])[None, :, :4], torch.tensor(d["fm_rigids0"])[None, :, 4:], psi)
    report["backbone_atoms"] = (maxrel(o37, a37), maxrel(o14, a14))
    print("backbone_atoms", report["backbone_atoms"], flush=True)
    np.savez_compressed(os.path.join(GOLD, "diffuser.npz"), **d)

    # ---------------- short reverse-diffusion trajectory (the loop of Experiment.inference_fn,
    # train_se3_diffusion.py:746-781, driven with the reference model + reference diffuser) ----------------
    import copy
    tconf = rl.base_conf(CACHE, num_blocks=2).model
    torch.manual_seed(0)
    tmodel = score_network.ScoreNetwork(tconf, diff)
    tmodel.load_state_dict(fo.synth_params(seed=21, conf=dict(fo.CONF, num_blocks=2)), strict=True)
    tmodel.eval()
    Bt, Nt, num_t, min_t, ns_ = 2, 10, 6, 0.01, 0.1
    np.random.seed(2024)
    zr = np.random.randn(Bt * Nt, 3); ur = np.random.rand(Bt * Nt); zt = np.random.normal(size=(Bt * Nt, 3))
    np.random.seed(2024)
    rig_init = diff.sample_ref(n_samples=Bt * Nt, as_tensor_7=True)["rigids_t"].reshape(Bt, Nt, 7)
    feats_t = dict(res_mask=torch.ones(Bt, Nt), fixed_mask=torch.zeros(Bt, Nt),
                   seq_idx=torch.arange(1, Nt + 1)[None].repeat(Bt, 1), torsion_angles_sin_cos=torch.zeros(Bt, Nt, 7, 2),
                   sc_ca_t=torch.zeros(Bt, Nt, 3), rigids_t=rig_init.clone(), t=torch.ones(Bt))
    steps = np.linspace(min_t, 1.0, num_t)[::-1]
    noises = []
    with torch.no_grad():
        feats_t["t"] = steps[0] * torch.ones(Bt)
        feats_t["sc_ca_t"] = tmodel(feats_t)["rigids"][..., 4:]
        for t_ in steps:
            if t_ > min_t:
                feats_t["t"] = t_ * torch.ones(Bt)
                mo = tmodel(feats_t)
                feats_t["sc_ca_t"] = mo["rigids"][..., 4:]
                st = np.random.get_state()
                z1 = np.random.normal(size=(Bt, Nt, 3)); z2 = np.random.normal(size=(Bt, Nt, 3))
                np.random.set_state(st)
                noises.append((z1, z2))
                rg = diff.reverse(rigid_t=ru.Rigid.from_tensor_7(feats_t["rigids_t"]), rot_score=du.move_to_np(mo["rot_score"]),
                                  trans_score=du.move_to_np(mo["trans_score"]), diffuse_mask=np.ones((Bt, Nt)), t=t_, dt=1 / num_t,
                                  center=True, noise_scale=ns_)
            else:
                mo = tmodel(feats_t)
                rg = ru.Rigid.from_tensor_7(mo["rigids"])
            feats_t["rigids_t"] = rg.to_tensor_7()
    np.savez_compressed(os.path.join(GOLD, "traj.npz"), B=Bt, N=Nt, num_t=num_t, min_t=min_t, noise_scale=ns_, seed=21,
                        blocks=2, init_randn=zr, init_rand=ur, init_normal=zt, rig_init=rig_init.numpy(),
                        z_rot=np.stack([n[0] for n in noises]), z_trans=np.stack([n[1] for n in noises]),
                        final_rigids=feats_t["rigids_t"].numpy(), final_psi=mo["psi"].numpy())
    print("trajectory golden written", flush=True)

    with open(os.path.join(GOLD, "PINNING_REPORT.txt"), "w") as f:
        f.write("oracle/framediff_oracle.py vs the unmodified reference (max |a-b| / max |b|)\n")
        f.write(f"torch {torch.__version__} numpy {np.__version__}\n")
        for k, v in report.items():
            f.write(f"{k}: {v}\n")
    print("wrote", GOLD)


if __name__ == "__main__":
    main()
