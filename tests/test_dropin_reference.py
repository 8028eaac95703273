"""The reference's OWN experiment code, unmodified, running on this repo's hot path.

Needs the reference checkout (build container only: /root/reference; skipped elsewhere).  Runs in a
subprocess because the reference's top-level module names (model, data, openfold, experiments) are
generic.  Inside: oracle/ref_loader stubs the reference's missing non-arithmetic imports (hydra, wandb,
...), se3_diffusion_amd.dropin.install() binds model.score_network / data.se3_diffuser / ... to the HIP
implementations (SIMT-interpreter build on this GPU-less box), then experiments.train_se3_diffusion.
Experiment is constructed and its loss_fn / update_fn / inference_fn are driven directly, then
experiments.inference_se3_diffusion.Sampler.sample on top of it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("FD_REFERENCE_ROOT", "/root/reference")

SCRIPT = r'''
import os, sys, types, copy
import numpy as np, torch
import json
ROOT, REF, MODE, OUT = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
from oracle import ref_loader as rl
rl.install()
DROPIN = MODE == "dropin"
bound = None
if DROPIN:
    import build_emu
    from se3_diffusion_amd import hip, dropin
    hip._TEST_OVERRIDE = hip.FdLib(build_emu.build(verbose=False))     # GPU-less box: interpreter build
    bound = dropin.install(REF)
from hydra.core.hydra_config import HydraConfig
HydraConfig.initialized = lambda: False
from experiments import train_se3_diffusion as tr
import model.score_network as sn, data.se3_diffuser as sd
if DROPIN:
    assert sn.__name__.startswith("se3_diffusion_amd") and sd.__name__.startswith("se3_diffusion_amd"), (sn.__name__, sd.__name__)
else:   # MODE == "reference": the same script on the reference's own modules -- the numbers the drop-in run is compared with
    assert sn.__name__ == "model.score_network" and sd.__name__ == "data.se3_diffuser" and sn.__file__.startswith(REF)
base = rl.base_conf(os.environ.get("FD_TEST_IGSO3_CACHE", "/tmp/fd_test_igso3_cache"), num_blocks=1)
ns = rl.ns
conf = ns(dict(
    data=dict(min_t=0.01, num_t=4, samples_per_eval_length=1, num_eval_lengths=1),
    experiment=dict(name="t", run_id=None, use_ddp=False, use_wandb=False, warm_start=None, use_warm_start_conf=False,
                    ckpt_dir=None, eval_dir=None, learning_rate=1e-4, num_parameters=None, batch_size=2,
                    trans_loss_weight=1.0, rot_loss_weight=0.5, rot_loss_t_threshold=0.2, separate_rot_loss=True,
                    trans_x0_threshold=1.0, coordinate_scaling=0.1, bb_atom_loss_weight=1.0, bb_atom_loss_t_filter=0.25,
                    dist_mat_loss_weight=1.0, dist_mat_loss_t_filter=0.25, aux_loss_weight=0.25, noise_scale=1.0)))
conf.diffuser, conf.model = base.diffuser, base.model
exp = tr.Experiment(conf=conf)
assert type(exp.model).__module__.startswith("se3_diffusion_amd") == DROPIN
from se3_diffusion_amd import train_step as ts
from oracle import framediff_oracle as fo
exp.model.load_state_dict(fo.synth_params(seed=5, conf=dict(fo.CONF, num_blocks=1)), strict=True)   # same weights in both modes
torch.manual_seed(0); np.random.seed(0)
B, N = 2, 8
batch = ts.synthetic_batch(B, N, "cpu", seed=3)
batch["t"][0] = 0.1
import random; random.seed(1)       # loss_fn draws random.random() for self-conditioning
exp.model.train()
loss, aux = exp.loss_fn(dict(batch))
assert torch.isfinite(loss), loss
random.seed(2)
l2, _ = exp.update_fn(dict(batch))     # fwd + loss + bwd + Adam through the reference's code
assert torch.isfinite(l2)
g = [p.grad for p in exp.model.parameters() if p.grad is not None]
assert len(g) > 70 and all(torch.isfinite(x).all() for x in g)
gnorm = {n: float(p.grad.double().norm()) for n, p in exp.model.named_parameters() if p.grad is not None}
pnorm_after = {n: float(p.detach().double().norm()) for n, p in exp.model.named_parameters()}
# reverse diffusion through the reference's inference_fn (diffuser.reverse on CPU-resident frames)
exp.model.eval()
np.random.seed(11)
init = exp.diffuser.sample_ref(n_samples=N, as_tensor_7=True)
feats = dict(res_mask=torch.ones(N), fixed_mask=torch.zeros(N), seq_idx=torch.arange(1, N + 1),
             torsion_angles_sin_cos=torch.zeros(N, 7, 2), sc_ca_t=torch.zeros(N, 3), rigids_t=init["rigids_t"])
feats = {k: v[None] for k, v in feats.items()}
out = exp.inference_fn(feats, num_t=3, min_t=0.01, aux_traj=True, noise_scale=0.1)
assert out["prot_traj"].shape == (3, 1, N, 37, 3) and np.isfinite(out["prot_traj"]).all()
# the reference's inference script: Sampler.sample (inference_se3_diffusion.py:418-459) on the same experiment.
# Sampler.__init__ (ESMFold download, output directories, checkpoint file) is bypassed; sample() only reads these.
for n in ("biotite.sequence", "biotite.sequence.io"):
    rl._stub(n)
from experiments import inference_se3_diffusion as inf
smp = object.__new__(inf.Sampler)
smp.exp, smp.diffuser, smp.device = exp, exp.diffuser, "cpu"
smp._diff_conf = ns(dict(num_t=3, min_t=0.01, noise_scale=0.1))
np.random.seed(12)
so = smp.sample(N)
assert so["prot_traj"].shape == (3, N, 37, 3) and np.isfinite(so["prot_traj"]).all(), so["prot_traj"].shape
assert so["rigid_traj"].shape[1:] == (N, 7) and np.isfinite(so["rigid_traj"]).all()
json.dump(dict(loss=float(loss), loss2=float(l2), aux={k: np.asarray(v.detach() if torch.is_tensor(v) else v, dtype=np.float64).reshape(-1).tolist()
                                                       for k, v in aux.items() if k in ("rot_loss", "trans_loss", "bb_atom_loss", "dist_mat_loss")},
               gnorm=gnorm, pnorm_after=pnorm_after, init=init["rigids_t"].numpy().tolist(),
               traj=np.asarray(out["prot_traj"], dtype=np.float64).tolist(), rigid_traj=np.asarray(out["rigid_traj"], dtype=np.float64).tolist(),
               sampler_traj=np.asarray(so["prot_traj"], dtype=np.float64).tolist()), open(OUT, "w"))
print("DROPIN_OK", float(loss), float(l2), bound)
'''


def _drive(tmp_path, mode):
    p = tmp_path / "drive.py"
    p.write_text(SCRIPT)
    out = tmp_path / f"{mode}.json"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, str(p), ROOT, REF, mode, str(out)], capture_output=True, text=True, cwd=str(tmp_path), env=env,
                       timeout=1500)
    assert "DROPIN_OK" in r.stdout, (mode, r.stdout[-2000:], r.stderr[-4000:])
    import json
    return json.load(open(out))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "experiments")), reason="reference checkout not present")
def test_reference_experiment_runs_on_dropin(tmp_path):
    """X1 with numbers: the SAME unmodified script (Experiment.loss_fn / update_fn / inference_fn, Sampler.sample) runs once on
    the reference's own modules and once on the drop-in (interpreter build), same weights, same seeded batch, same numpy /
    python RNG streams; loss, per-term losses, every parameter-gradient norm, the parameters after the reference's Adam step and
    the sampled trajectories must agree (fp32 tolerances of DESIGN.md "Numerics"; Angstrom for coordinates)."""
    import numpy as np
    got = _drive(tmp_path, "dropin")
    ref = _drive(tmp_path, "reference")
    assert abs(got["loss"] - ref["loss"]) < 1e-4 * abs(ref["loss"]) + 1e-6, (got["loss"], ref["loss"])
    assert abs(got["loss2"] - ref["loss2"]) < 1e-4 * abs(ref["loss2"]) + 1e-6
    for k, v in ref["aux"].items():
        assert np.allclose(got["aux"][k], v, rtol=2e-4, atol=1e-5), k
    # the parameters that never receive a gradient in the reference (linear_rbf, torsion_pred.linear_3: .grad stays None) get an
    # exactly-zero gradient here (the one autograd node returns a tensor for every parameter): Adam leaves them unchanged
    extra = set(got["gnorm"]) - set(ref["gnorm"])
    assert set(ref["gnorm"]) <= set(got["gnorm"]) and all(got["gnorm"][n] == 0.0 for n in extra), extra
    assert all(("linear_rbf" in n) or ("torsion_pred.linear_3" in n) for n in extra), extra
    gmax = max(ref["gnorm"].values())
    for n, v in ref["gnorm"].items():
        assert abs(got["gnorm"][n] - v) < 2e-3 * v + 2e-5 * gmax, (n, got["gnorm"][n], v)
    # after the reference's Adam step.  Adam's first update is lr * sign(g): a parameter whose gradient is analytically zero
    # (linear_b.bias: softmax shift invariance) moves by +-lr per entry with the sign of fp32 round-off -- skipped.
    for n, v in ref["pnorm_after"].items():
        if ref["gnorm"].get(n, 0.0) < 1e-5 * gmax:
            continue
        assert abs(got["pnorm_after"][n] - v) < 1e-4 * v + 1e-6, (n, got["pnorm_after"][n], v, ref["gnorm"].get(n))
    a, b = np.array(got["init"]), np.array(ref["init"])                            # sample_ref from the same numpy stream
    assert np.abs(a[..., 4:] - b[..., 4:]).max() < 1e-4 and np.abs(np.abs((a[..., :4] * b[..., :4]).sum(-1)) - 1).max() < 1e-5
    for k, tol in (("traj", 2e-2), ("sampler_traj", 2e-2)):
        a, b = np.array(got[k]), np.array(ref[k])
        assert a.shape == b.shape and np.abs(a - b).max() < tol, (k, float(np.abs(a - b).max()))
    a, b = np.array(got["rigid_traj"]), np.array(ref["rigid_traj"])
    assert np.abs(a[..., 4:] - b[..., 4:]).max() < 2e-2
    assert np.abs(np.abs((a[..., :4] * b[..., :4]).sum(-1)) - 1).max() < 1e-4    # quaternions up to sign
