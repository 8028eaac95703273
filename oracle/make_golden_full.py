"""TEST INFRASTRUCTURE -- golden fixtures at the sizes the product SHIPS at (BASELINE.json configs 2/3/5), generated
from the UNMODIFIED reference (model/score_network.py:170-215, experiments/train_se3_diffusion.py:746-781):

    fwd_n128_b2   full depth (4 blocks), B=2 x N=128: outputs + gradient signatures of all 282 parameters
    fwd_n256_b1   full depth, B=1 x N=256: outputs + gradient signatures
    fwd_n512_b1   full depth, B=1 x N=512: outputs + gradient signatures (BASELINE configs[3] trains at N up to 512)
    traj_n128     5 reverse-diffusion steps of Experiment.inference_fn's loop at N=128, full depth, injected noise
    traj_n256     BASELINE configs[2]'s length: 5 reverse steps at B=1 x N=256  } produced by calling the UNMODIFIED
    traj_n512_b2  BASELINE configs[4]'s length: 5 reverse steps at B=2 x N=512  } Experiment.inference_fn itself (a recording
                  wrapper around diffuser.reverse snapshots the numpy RNG state to learn the draws it is about to make)
    traj_n128_t50 50 reverse steps (51 forwards) of the unmodified Experiment.inference_fn at B=1 x N=128: error growth of the
                  device-resident / hipGraph-replayed loop over a tenth of the metric's 500-step trajectory
    traj_n128_t500  THE METRIC'S SCHEDULE: 500 reverse steps (501 forwards, dt = 1/500, the 500-point t grid of
                  config/inference.yaml) of the unmodified Experiment.inference_fn at B=1 x N=128; every 10th step's frames are
                  stored, and instead of the 3 MB of normal draws the state of numpy's global generator at the first
                  diffuser.reverse call (the generator checks that the draws of all 500 calls are consecutive from it)
    traj_n256_t500  BASELINE.json configs[2] exactly ("500 steps, N=256"): the same at B=1 x N=256 (631 s of reference CPU time)

Run in the build container only (needs /root/reference):  python oracle/make_golden_full.py
The oracle (oracle/framediff_oracle.py) is pinned against the same runs (PINNING_REPORT_FULL.txt).
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
from oracle.make_golden import maxrel, quat_sign_align  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CACHE = os.environ.get("FD_IGSO3_CACHE", "/tmp/fd_igso3_cache")
OUT_KEYS = ["rot_score", "trans_score", "rigids", "atom37", "psi"]


def main():
    rl.install()
    from data import se3_diffuser, utils as du  # reference modules
    from model import score_network
    from openfold.utils import rigid_utils as ru

    torch.set_num_threads(int(os.environ.get("FD_GOLDEN_THREADS", "8")))
    conf = rl.base_conf(CACHE)
    diff = se3_diffuser.SE3Diffuser(conf.diffuser)
    report = {}
    cases = [
        dict(name="fwd_n128_b2", B=2, N=128, seed=31, n_pad=3, n_fixed=2, blocks=4, grad=True),
        dict(name="fwd_n256_b1", B=1, N=256, seed=32, n_pad=0, n_fixed=0, blocks=4, grad=True),
        dict(name="fwd_n512_b1", B=1, N=512, seed=33, n_pad=0, n_fixed=0, blocks=4, grad=True),
    ]
    only = os.environ.get("FD_GOLDEN_ONLY")
    for c in cases:
        if only and c["name"] not in only.split(","):
            continue
        mconf = rl.base_conf(CACHE, num_blocks=c["blocks"]).model
        oconf = dict(fo.CONF, num_blocks=c["blocks"])
        model = score_network.ScoreNetwork(mconf, diff)
        P = fo.synth_params(seed=c["seed"], conf=oconf)
        model.load_state_dict(P, strict=True)
        feats = fo.synth_feats(c["B"], c["N"], seed=c["seed"], n_pad=c["n_pad"], n_fixed=c["n_fixed"])
        save = dict(B=c["B"], N=c["N"], seed=c["seed"], n_pad=c["n_pad"], n_fixed=c["n_fixed"], blocks=c["blocks"])
        rep = {}
        model.train()
        t0 = time.time()
        if c["grad"]:
            out = model({k: v.clone() for k, v in feats.items()})
            rs = np.random.RandomState(77 + c["seed"])
            wts = {k: torch.tensor(rs.standard_normal(tuple(out[k].shape))).to(out[k].dtype) for k in OUT_KEYS}
            loss = sum((out[k] * wts[k]).sum() for k in wts)
            loss.backward()
            rep["t_ref_s"] = time.time() - t0
            grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            t0 = time.time()
            oo = fo.score_network_forward(Po, feats, oconf, tfmr_mask_mode="additive")
            lo = sum((oo[k] * wts[k]).sum() for k in wts)
            lo.backward()
            rep["t_oracle_s"] = time.time() - t0
            gerr = {}
            for n, g in grads.items():
                go = Po[n].grad
                gerr[n] = float((go.double() - g.double()).abs().max() / (g.double().abs().max() + 1e-12)) if go is not None else 1.0
            rep["grad_worst"] = sorted(gerr.items(), key=lambda kv: -kv[1])[:6]
            rep["loss"] = (float(loss), float(lo))
            save["w_seed"] = 77 + c["seed"]              # the loss weights are regenerated from the seed (RandomState stream)
            save["loss"] = float(loss)
            for n, g in grads.items():
                if g.numel() <= 512:
                    save["grad/" + n] = g.numpy()
                else:
                    save["gsig/" + n] = np.array([g.double().sum(), g.double().abs().sum(), g.double().norm()])
        else:
            with torch.enable_grad():
                out = model({k: v.clone() for k, v in feats.items()})   # train mode + grad enabled: additive tfmr mask
            rep["t_ref_s"] = time.time() - t0
            with torch.no_grad():
                oo = fo.score_network_forward(P, feats, oconf, tfmr_mask_mode="additive")
        for k in ["psi", "rot_score", "trans_score", "atom37", "atom14"]:
            rep[k] = maxrel(oo[k].detach(), out[k].detach())
        rep["rigids"] = maxrel(quat_sign_align(oo["rigids"].detach(), out["rigids"].detach()), out["rigids"].detach())
        for k, v in out.items():
            a = v.detach().numpy()
            if k in ("atom37", "atom14"):
                a = a[:, :, :5]                                          # only the backbone slots are non-zero
            save["out_" + k] = a
        report[c["name"]] = rep
        print(c["name"], {k: (v if not isinstance(v, float) else f"{v:.2e}") for k, v in rep.items()}, flush=True)
        np.savez_compressed(os.path.join(GOLD, c["name"] + ".npz"), **save)

    # ---------------- trajectory at N=128, full depth, injected noise ----------------
    if not only or "traj_n128" in only.split(","):
        tconf = rl.base_conf(CACHE, num_blocks=4).model
        tmodel = score_network.ScoreNetwork(tconf, diff)
        tmodel.load_state_dict(fo.synth_params(seed=41, conf=dict(fo.CONF, num_blocks=4)), strict=True)
        tmodel.eval()
        Bt, Nt, num_t, min_t, ns_ = 1, 128, 5, 0.01, 0.1
        np.random.seed(4242)
        zr = np.random.randn(Bt * Nt, 3); ur = np.random.rand(Bt * Nt); zt = np.random.normal(size=(Bt * Nt, 3))
        np.random.seed(4242)
        rig_init = diff.sample_ref(n_samples=Bt * Nt, as_tensor_7=True)["rigids_t"].reshape(Bt, Nt, 7)
        feats_t = dict(res_mask=torch.ones(Bt, Nt), fixed_mask=torch.zeros(Bt, Nt),
                       seq_idx=torch.arange(1, Nt + 1)[None].repeat(Bt, 1), torsion_angles_sin_cos=torch.zeros(Bt, Nt, 7, 2),
                       sc_ca_t=torch.zeros(Bt, Nt, 3), rigids_t=rig_init.clone(), t=torch.ones(Bt))
        steps = np.linspace(min_t, 1.0, num_t)[::-1]
        noises, per_step = [], []
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
                                      trans_score=du.move_to_np(mo["trans_score"]), diffuse_mask=np.ones((Bt, Nt)), t=t_,
                                      dt=1 / num_t, center=True, noise_scale=ns_)
                else:
                    mo = tmodel(feats_t)
                    rg = ru.Rigid.from_tensor_7(mo["rigids"])
                feats_t["rigids_t"] = rg.to_tensor_7()
                per_step.append(feats_t["rigids_t"].numpy().copy())
        np.savez_compressed(os.path.join(GOLD, "traj_n128.npz"), B=Bt, N=Nt, num_t=num_t, min_t=min_t, noise_scale=ns_, seed=41,
                            blocks=4, init_randn=zr, init_rand=ur, init_normal=zt, rig_init=rig_init.numpy(),
                            z_rot=np.stack([n[0] for n in noises]), z_trans=np.stack([n[1] for n in noises]),
                            final_rigids=feats_t["rigids_t"].numpy(), final_psi=mo["psi"].numpy(), step_rigids=np.stack(per_step))
        print("trajectory golden (N=128) written", flush=True)

    for name, Bt, Nt, num_t, seed, stride in (("traj_n256", 1, 256, 5, 42, 1), ("traj_n512_b2", 2, 512, 5, 43, 1),
                                              ("traj_n128_t50", 1, 128, 50, 44, 1), ("traj_n128_t500", 1, 128, 500, 45, 10),
                                              ("traj_n256_t500", 1, 256, 500, 46, 10)):
        if not only or name in only.split(","):
            t0 = time.time()
            traj_via_experiment(name, Bt, Nt, num_t, seed, stride=stride)
            print(f"{name} written ({time.time() - t0:.0f} s)", flush=True)

    if not report:
        return
    with open(os.path.join(GOLD, "PINNING_REPORT_FULL.txt"), "a" if only else "w") as f:
        f.write("oracle/framediff_oracle.py vs the unmodified reference at shipped sizes (max |a-b| / max |b|)\n")
        f.write(f"torch {torch.__version__} numpy {np.__version__}\n")
        for k, v in report.items():
            f.write(f"{k}: {v}\n")
    print("wrote", GOLD)


def traj_via_experiment(name, B, N, num_t, seed, min_t=0.01, noise_scale=0.1, stride=1):
    """The reverse loop of the reference's own Experiment.inference_fn (experiments/train_se3_diffusion.py:718-818), called
    unmodified on the reference's ScoreNetwork / SE3Diffuser.  Recorded: the initial draws of sample_ref, the (rot, trans)
    normal draws of every diffuser.reverse call (se3_diffuser.py:213-262: rotation first), every step's frames, the last psi.
    stride > 1 (the 500-step fixture): every stride-th step's frames, and the generator STATE at the first reverse call instead of
    the draws themselves (legacy numpy streams are stable across versions; checked here: all draws are consecutive from it)."""
    from hydra.core.hydra_config import HydraConfig
    HydraConfig.initialized = lambda: False
    from experiments import train_se3_diffusion as tr
    base = rl.base_conf(CACHE, num_blocks=4)
    conf = rl.ns(dict(
        data=dict(min_t=min_t, num_t=num_t, samples_per_eval_length=1, num_eval_lengths=1),
        experiment=dict(name="golden", run_id=None, use_ddp=False, use_wandb=False, warm_start=None, use_warm_start_conf=False,
                        ckpt_dir=None, eval_dir=None, learning_rate=1e-4, num_parameters=None, batch_size=B,
                        trans_loss_weight=1.0, rot_loss_weight=0.5, rot_loss_t_threshold=0.2, separate_rot_loss=True,
                        trans_x0_threshold=1.0, coordinate_scaling=0.1, bb_atom_loss_weight=1.0, bb_atom_loss_t_filter=0.25,
                        dist_mat_loss_weight=1.0, dist_mat_loss_t_filter=0.25, aux_loss_weight=0.25, noise_scale=1.0)))
    conf.diffuser, conf.model = base.diffuser, base.model
    exp = tr.Experiment(conf=conf)
    assert type(exp.model).__module__ == "model.score_network" and type(exp.diffuser).__module__ == "data.se3_diffuser"
    exp.model.load_state_dict(fo.synth_params(seed=seed, conf=dict(fo.CONF, num_blocks=4)), strict=True)
    exp.model.eval()
    np.random.seed(1000 + seed)
    zr = np.random.randn(B * N, 3); ur = np.random.rand(B * N); zt = np.random.normal(size=(B * N, 3))
    np.random.seed(1000 + seed)
    rig_init = exp.diffuser.sample_ref(n_samples=B * N, as_tensor_7=True)["rigids_t"].reshape(B, N, 7)
    feats = dict(res_mask=torch.ones(B, N), fixed_mask=torch.zeros(B, N), seq_idx=torch.arange(1, N + 1)[None].repeat(B, 1),
                 torsion_angles_sin_cos=torch.zeros(B, N, 7, 2), sc_ca_t=torch.zeros(B, N, 3), rigids_t=rig_init.clone())
    noises = []
    orig_reverse = exp.diffuser.reverse

    first_state = []

    def recording_reverse(*a, **kw):
        st = np.random.get_state()
        if not first_state:
            first_state.append(st)
        noises.append((np.random.normal(size=(B, N, 3)), np.random.normal(size=(B, N, 3))))
        np.random.set_state(st)
        return orig_reverse(*a, **kw)
    exp.diffuser.reverse = recording_reverse
    out = exp.inference_fn(feats, num_t=num_t, min_t=min_t, aux_traj=True, noise_scale=noise_scale)
    rt = np.asarray(out["rigid_traj"])[::-1]            # inference_fn flips the trajectory: back to init, step 1, ...
    assert rt.shape == (num_t + 1, B, N, 7) and np.abs(rt[0] - rig_init.numpy()).max() == 0.0
    if stride > 1:
        # nothing but diffuser.reverse consumed the global generator between the calls: the recorded draws are one stream
        rs = np.random.RandomState()
        rs.set_state(first_state[0])
        for zr_, zt_ in noises:
            assert np.array_equal(rs.normal(size=(B, N, 3)), zr_) and np.array_equal(rs.normal(size=(B, N, 3)), zt_)
        kind, keys, pos, has_gauss, cached = first_state[0]
        idx = np.arange(stride - 1, num_t, stride)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), B=B, N=N, num_t=num_t, min_t=min_t, noise_scale=noise_scale,
                            seed=seed, blocks=4, init_randn=zr, init_rand=ur, init_normal=zt, rig_init=rig_init.numpy(),
                            rng_keys=np.asarray(keys, dtype=np.uint32), rng_pos=int(pos), rng_has_gauss=int(has_gauss),
                            rng_cached=float(cached), n_reverse=len(noises), step_index=idx,
                            final_rigids=rt[-1].astype(np.float32), final_psi=out["psi_pred"][0].numpy(),
                            step_rigids=rt[1:][idx].astype(np.float32))
        return
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), B=B, N=N, num_t=num_t, min_t=min_t, noise_scale=noise_scale, seed=seed,
                        blocks=4, init_randn=zr, init_rand=ur, init_normal=zt, rig_init=rig_init.numpy(),
                        z_rot=np.stack([n[0] for n in noises]), z_trans=np.stack([n[1] for n in noises]),
                        final_rigids=rt[-1].astype(np.float32), final_psi=out["psi_pred"][0].numpy(),
                        step_rigids=rt[1:].astype(np.float32))


if __name__ == "__main__":
    main()
