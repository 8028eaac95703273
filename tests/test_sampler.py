"""Reverse-diffusion trajectory parity: se3_diffusion_amd.sampler.sample (device-resident loop) vs a
trajectory produced by the UNMODIFIED reference model + diffuser following Experiment.inference_fn
(tests/golden/traj.npz, oracle/make_golden.py), with the reference's numpy draws injected."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import framediff_oracle as fo  # noqa: E402
from se3_diffusion_amd import sampler, train_step as ts  # noqa: E402
from se3_diffusion_amd.data import utils as du  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402
from test_diffuser import conf as dconf  # noqa: E402
from se3_diffusion_amd.data import se3_diffuser  # noqa: E402

T = np.load(os.path.join(ROOT, "tests", "golden", "traj.npz"))


def _run(dev):
    diff = se3_diffuser.SE3Diffuser(dconf())
    blocks = int(T["blocks"])
    m = ScoreNetwork(ts.base_model_conf(blocks), diff)
    m.load_state_dict(fo.synth_params(seed=int(T["seed"]), conf=dict(fo.CONF, num_blocks=blocks)), strict=True)
    m = m.to(dev).eval()
    B, N = int(T["B"]), int(T["N"])
    feats = sampler.init_feats(diff, B, N, dev, noise=(T["init_randn"], T["init_rand"], T["init_normal"]))
    ref0 = T["rig_init"]
    assert np.abs(du.quat_wxyz_to_matrix(feats["rigids_t"].cpu().numpy()[..., :4].astype(np.float64))
                  - du.quat_wxyz_to_matrix(ref0[..., :4].astype(np.float64))).max() < 3e-6
    zr, zt = T["z_rot"], T["z_trans"]
    out = sampler.sample(m, diff, feats, num_t=int(T["num_t"]), min_t=float(T["min_t"]), noise_scale=float(T["noise_scale"]),
                         noise_fn=lambda i, shp: (zr[i], zt[i]))
    got = out["rigids"].cpu().numpy()
    ref = T["final_rigids"]
    rm = lambda q: du.quat_wxyz_to_matrix(q[..., :4].astype(np.float64))
    assert np.abs(rm(got) - rm(ref)).max() < 5e-4
    assert np.abs(got[..., 4:] - ref[..., 4:]).max() < 5e-3      # Angstrom, after 6 chained network calls
    assert np.abs(out["psi"].cpu().numpy() - T["final_psi"]).max() < 2e-3
    assert out["atom37"].shape == (B, N, 37, 3)


def test_trajectory_emu(use_emu):
    _run("cpu")


@pytest.mark.gpu
def test_trajectory_gpu(hip_lib):
    _run("cuda")


@pytest.mark.gpu
def test_graph_fork_matches_single_stream_gpu(hip_lib):
    """options.graph_fork (independent launches of the sampling forward on a second stream = parallel branches of the captured
    hipGraph; off by default, measured slower) gives the trajectory of the single-stream forward: same noise, eager and graph."""
    from se3_diffusion_amd import options
    dev = "cuda"
    diff = se3_diffuser.SE3Diffuser(dconf())
    m = ScoreNetwork(ts.base_model_conf(2), diff)
    m.load_state_dict(fo.synth_params(seed=3, conf=dict(fo.CONF, num_blocks=2)), strict=True)
    m = m.to(dev).eval()
    B, N, steps = 1, 64, 6
    g = torch.Generator().manual_seed(0)
    z = [(torch.randn(B, N, 3, generator=g, dtype=torch.float64), torch.randn(B, N, 3, generator=g, dtype=torch.float64))
         for _ in range(steps)]
    outs = []
    for fork, graph in ((False, False), (True, False), (True, True)):
        with options.override(graph_fork=fork):
            feats = sampler.init_feats(diff, B, N, dev, generator=torch.Generator(device=dev).manual_seed(1))
            outs.append(sampler.sample(m, diff, feats, num_t=steps, noise_fn=lambda i, shape: z[i], use_graph=graph)["rigids"].clone())
    for o in outs[1:]:
        assert float((o - outs[0]).abs().max()) < 1e-4
