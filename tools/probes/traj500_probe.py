import os, sys
import numpy as np, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
from oracle import framediff_oracle as fo
from se3_diffusion_amd import sampler, train_step as ts
from se3_diffusion_amd.data import se3_diffuser, utils as du
from se3_diffusion_amd.model.score_network import ScoreNetwork
from test_diffuser import conf as dconf
T = np.load(ROOT+'/tests/golden/traj_n128_t500.npz')
diff = se3_diffuser.SE3Diffuser(dconf())
m = ScoreNetwork(ts.base_model_conf(4), diff)
m.load_state_dict(fo.synth_params(seed=int(T["seed"]), conf=dict(fo.CONF, num_blocks=4)), strict=True)
m = m.cuda().eval()
B, N = 1, 128
feats = sampler.init_feats(diff, B, N, "cuda", noise=(T["init_randn"], T["init_rand"], T["init_normal"]))
rs = np.random.RandomState(); rs.set_state(("MT19937", T["rng_keys"], int(T["rng_pos"]), int(T["rng_has_gauss"]), float(T["rng_cached"])))
out = sampler.sample(m, diff, feats, num_t=500, min_t=0.01, noise_scale=0.1, noise_fn=lambda i, shp: (rs.normal(size=shp), rs.normal(size=shp)), return_traj=True, use_graph=True)
rm = lambda q: du.quat_wxyz_to_matrix(np.asarray(q)[..., :4].astype(np.float64))
idx = [int(i) for i in T["step_index"]]
for k in (-3, -2, -1):
    got = out["rigid_traj"][idx[k]].cpu().numpy(); ref = T["step_rigids"][k]
    er = np.abs(rm(got) - rm(ref)).max(axis=(-1, -2))[0]; et = np.abs(got[..., 4:] - ref[..., 4:]).max(-1)[0]
    print("step", idx[k] + 1, "rot err: max %.2e median %.2e 90%% %.2e #>1e-4: %d argmax %d | trans max %.2e median %.2e" % (er.max(), np.median(er), np.quantile(er, .9), int((er > 1e-4).sum()), int(er.argmax()), et.max(), np.median(et)))
# the last step is the network's own frame prediction: how sensitive is it?  same forward on the REFERENCE's step-499 state is not available (every 10th stored);
# perturb the input frames of the last forward by 1e-5 and look at the output change
# the last step of the loop is the NETWORK'S OWN frame prediction (train_se3_diffusion.py:778-780), not a reverse step.  How much
# does that forward amplify a difference of its input frames?  Input = the reference's step-490 frames, perturbed by 1e-5.
base = torch.tensor(T["step_rigids"][-2]).cuda()
g = torch.Generator(device="cuda").manual_seed(3)
f0 = sampler.init_feats(diff, B, N, "cuda", generator=g)
f0["t"] = torch.full((B,), 0.03, device="cuda")
f0["sc_ca_t"] = base[..., 4:].clone()
def fwd(r):
    f = dict(f0, rigids_t=r)
    with torch.no_grad():
        return m(f)["rigids"].cpu().numpy()
r0 = fwd(base)
for eps in (1e-5, 1e-4):
    d = torch.randn(base.shape, device="cuda", generator=g) * eps
    r1 = fwd(base + d)
    er = np.abs(rm(r1) - rm(r0)).max(axis=(-1, -2))[0]
    print("step-like forward: input perturbed by %.0e -> output rot change max %.2e median %.2e (amplification of the max: %.0fx)" % (eps, er.max(), np.median(er), er.max() / eps))
