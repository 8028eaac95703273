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


class _Opaque(torch.nn.Module):
    """a model the sampler cannot see through (no .module, not a ScoreNetwork): its self-conditioning input has to be updated by
    the loop's explicit copy, not by the heads kernel (the kernel-side write was the only path)"""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self._model_conf = net._model_conf

    def forward(self, feats):
        return self.net(feats)


class _Wrapped(torch.nn.Module):
    """DDP-style wrapper: the sampler unwraps .module and lets the heads kernel write sc_ca_t"""

    def __init__(self, net):
        super().__init__()
        self.module = net
        self._model_conf = net._model_conf

    def forward(self, feats):
        return self.module(feats)


def _run(dev, wrap=None):
    diff = se3_diffuser.SE3Diffuser(dconf())
    blocks = int(T["blocks"])
    m = ScoreNetwork(ts.base_model_conf(blocks), diff)
    m.load_state_dict(fo.synth_params(seed=int(T["seed"]), conf=dict(fo.CONF, num_blocks=blocks)), strict=True)
    m = m.to(dev).eval()
    if wrap is not None:
        m = wrap(m)
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


@pytest.mark.parametrize("wrap", [_Opaque, _Wrapped])
def test_trajectory_wrapped_model_emu(use_emu, wrap):
    """the same trajectory through a model wrapper: self-conditioning must still be updated every step"""
    _run("cpu", wrap)


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


def _advance(lib, dev):
    """fd_sample_advance: the first node of a captured diffusion step -- t, the reverse step's scalars and the step's draws from
    device arrays indexed by a device counter, counter += 1 (experiments/train_se3_diffusion.py:746-781: the host side of the
    reference's loop)"""
    K, B, nz, steps = 3, 5, 2 * 5 * 7 * 3, 7
    g = torch.Generator().manual_seed(2)
    all_t = torch.rand(steps, generator=g).to(dev)
    all_tp = torch.rand(steps, 2, generator=g, dtype=torch.float64).to(dev)
    z_all = torch.randn(K, nz, generator=g, dtype=torch.float64).to(dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    t_out, tp, z = torch.zeros(B, device=dev), torch.zeros(2, dtype=torch.float64, device=dev), torch.zeros(nz, dtype=torch.float64, device=dev)
    for i in range(steps):
        lib.call("fd_sample_advance", counter, all_t, all_tp, z_all, K, nz, t_out, B, tp, z)
        assert int(counter.item()) == i + 1
        assert torch.equal(t_out, all_t[i].expand(B)) and torch.equal(tp, all_tp[i]) and torch.equal(z, z_all[i % K])


def test_sample_advance_emu(emu_lib):
    _advance(emu_lib, "cpu")


@pytest.mark.gpu
def test_device_steps_match_host_steps_gpu(hip_lib):
    """the captured step with fd_sample_advance as its first node (options.sampler_device_steps: no launch between graph replays)
    against the same graph behind the three host-issued launches per step (fill_, copy_, copy_ / normal_) and against the eager
    loop: same injected noise, bit-identical frames.  12 steps with NOISE_STEPS = 50 and 7 steps per refill are both covered by
    patching the refill period."""
    from se3_diffusion_amd import options
    _advance(hip_lib, "cuda")
    dev = "cuda"
    diff = se3_diffuser.SE3Diffuser(dconf())
    m = ScoreNetwork(ts.base_model_conf(2), diff)
    m.load_state_dict(fo.synth_params(seed=5, conf=dict(fo.CONF, num_blocks=2)), strict=True)
    m = m.to(dev).eval()
    B, N, steps = 2, 48, 12
    g = torch.Generator().manual_seed(0)
    z = [(torch.randn(B, N, 3, generator=g, dtype=torch.float64), torch.randn(B, N, 3, generator=g, dtype=torch.float64))
         for _ in range(steps)]
    outs = []
    for dev_steps, graph in ((False, False), (False, True), (True, True)):
        with options.override(sampler_device_steps=dev_steps):
            feats = sampler.init_feats(diff, B, N, dev, generator=torch.Generator(device=dev).manual_seed(1))
            r = sampler.sample(m, diff, feats, num_t=steps, noise_fn=lambda i, shape: z[i], use_graph=graph, return_traj=True)
            outs.append(torch.stack(r["rigid_traj"]).clone())
    assert torch.equal(outs[1], outs[2])
    assert float((outs[0] - outs[1]).abs().max()) < 1e-5
    # device-drawn noise: a seeded generator gives the same trajectory twice, and finite frames
    res = []
    for _ in range(2):
        gen = torch.Generator(device=dev).manual_seed(77)
        feats = sampler.init_feats(diff, B, N, dev, generator=gen)
        res.append(sampler.sample(m, diff, feats, num_t=60, generator=gen, use_graph=True)["rigids"].clone())
    assert torch.equal(res[0], res[1]) and torch.isfinite(res[0]).all()


@pytest.mark.gpu
def test_kernels_per_captured_step_gpu(hip_lib):
    """the library's launches inside ONE captured diffusion step of a lone backbone on the full-depth model (fd_launch_count across the
    capture; bench.py reports it as config.sampling.*.kernels_per_step): 130 at the end of round 6 -- a launch added to the sampling
    forward shows up here, not as a slower benchmark"""
    dev = "cuda"
    diff = se3_diffuser.SE3Diffuser(dconf())
    m = ScoreNetwork(ts.base_model_conf(4), diff).to(dev).eval()
    st = {}
    feats = sampler.init_feats(diff, 1, 128, dev, generator=torch.Generator(device=dev).manual_seed(1))
    sampler.sample(m, diff, feats, num_t=8, min_t=0.01, noise_scale=0.1, generator=torch.Generator(device=dev).manual_seed(2), use_graph=True,
                   stats=st)
    assert 100 <= st["kernels_per_step"] <= 132, st
