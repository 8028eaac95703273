"""Device-resident reverse diffusion: the loop of Experiment.inference_fn
(reference experiments/train_se3_diffusion.py:718-818) with the frames kept in HBM for all
num_t steps -- no device->numpy->scipy->eigh->device round trip per step
(reference :770-781, se3_diffuser.py:11-29) and B >= 1 equal-length backbones per launch.

Per step: ScoreNetwork forward (HIP trunk) -> self-conditioning CA update -> fd_se3_reverse_step.
The last step (t == min_t) takes the model's predicted frames directly (:778-780).

use_graph=True captures ONE step (~230 kernel launches) into a hipGraph and replays it for every t:
the time-dependent scalars live in HBM (feats['t'], tparams = (g_rot(t), b(t))) and the noise is drawn
into static buffers before each replay, so small-N sampling is no longer launch-bound.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip
from .options import opts


def init_feats(diffuser, B, N, device, generator=None, noise=None):
    """Sampler.sample's data_init (inference_se3_diffusion.py:427-449) for B backbones of length N."""
    rig = diffuser.sample_ref_device(B * N, device, noise=noise, generator=generator).view(B, N, 7)
    dev = torch.device(device)
    return dict(
        res_mask=torch.ones(B, N, device=dev), fixed_mask=torch.zeros(B, N, device=dev),
        seq_idx=torch.arange(1, N + 1, device=dev)[None].repeat(B, 1),
        torsion_angles_sin_cos=torch.zeros(B, N, 7, 2, device=dev), sc_ca_t=torch.zeros(B, N, 3, device=dev),
        rigids_t=rig, t=torch.ones(B, device=dev))


def _f64(x, dev):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(device=dev, dtype=torch.float64)


@torch.no_grad()
def sample(model, diffuser, feats, num_t=500, min_t=0.01, noise_scale=0.1, self_condition=True, center=True,
           generator=None, noise_fn=None, return_traj=False, use_graph=False, stats=None):
    """Run the reverse process on `feats` (from init_feats).  noise_fn(step, shape) -> (z_rot, z_trans) injects
    draws (e.g. the numpy stream, for trajectory parity); default draws on the device.
    Returns dict(rigids [B,N,7], atom37 [B,N,37,3], psi, (rigid_traj list)).  `stats` (a dict) receives loop_ms = device
    time of the reverse loop itself (HIP events; excludes the one-off warm-up + graph capture) and its step count.

    Random stream: with device-drawn noise the graph path (options.sampler_device_steps) fills the draws of 50 steps with ONE
    normal_ launch on a [50, 2, B, N, 3] buffer, the eager path draws [2, B, N, 3] per step: for one generator seed the two paths
    consume the Philox stream differently and give different (equally distributed) trajectories.  Injected noise (noise_fn) is
    consumed in step order on both paths -- that is what the trajectory parity tests compare."""
    dev = feats["rigids_t"].device
    B, N = feats["res_mask"].shape
    steps = np.linspace(min_t, 1.0, num_t)[::-1]
    dt = 1.0 / num_t
    was_training = model.training
    model.eval()
    # masks and weights are constant over the trajectory: their derived tensors (pair mask, diffuse mask, key mask,
    # packed linear_b / down_z weights) are built by the first forward and reused by the other 500 (trunk._cached)
    model._fd_static = {}
    lib = hip.get_lib()
    from . import ops as _ops0
    _ops0.edge_sched_init(dev)       # (the tile-counter pool of the fused edge kernels exists before anything is captured)
    saved_prof, lib.gemm_profile = lib.gemm_profile, (None if use_graph else lib.gemm_profile)
    # static state (updated in place so a captured graph sees it)
    st = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in feats.items()}
    st["rigids_t"] = st["rigids_t"].to(torch.float32).contiguous()
    st["t"] = torch.ones(B, device=dev)
    diffuse_mask = ((1 - st["fixed_mask"]) * st["res_mask"]).contiguous()
    z_both = torch.zeros((2, B, N, 3), dtype=torch.float64, device=dev)      # (one normal_ launch fills both draws: rot, then trans)
    z_rot, z_trans = z_both[0], z_both[1]
    tparams = torch.zeros(2, dtype=torch.float64, device=dev)
    psi = torch.zeros((B, N, 2), device=dev)
    so3, r3 = diffuser._so3_diffuser, diffuser._r3_diffuser

    # per-step scalars (g_rot(t), b(t)) uploaded once; each step copies its row device-to-device
    all_tp = torch.tensor(np.stack([[float(so3.diffusion_coef(t)), float(r3.b_t(t))] for t in steps]),
                          dtype=torch.float64).to(dev)
    step_of = {float(t): i for i, t in enumerate(steps)}

    def set_t(t):
        st["t"].fill_(float(t))
        tparams.copy_(all_tp[step_of[float(t)]])

    def draw(i):
        if noise_fn is None:
            z_both.normal_(generator=generator)
        else:
            zr, zt = noise_fn(i, (B, N, 3))
            z_rot.copy_(_f64(zr, dev))
            z_trans.copy_(_f64(zt, dev))

    # reference quirk kept (train_se3_diffusion.py:753-756 vs :763-765): `self_condition` gates only the initial extra
    # forward; the per-step sc_ca_t update depends on the MODEL's embed_self_conditioning flag alone
    embed_sc = bool(getattr(getattr(getattr(model, "_model_conf", None), "embed", None), "embed_self_conditioning", True))
    # The heads kernel of THIS package's ScoreNetwork can write the predicted CA positions straight into st["sc_ca_t"]
    # (module attribute _fd_sc_ca_out, read by _ScoreNetFn.forward); any other callable -- a wrapper (DDP, a stub in a test) that
    # does not reach that code -- gets the explicit copy after its forward.  The first forward checks that the kernel really wrote.
    from .model.score_network import ScoreNetwork as _SN
    inner = model
    while not isinstance(inner, _SN) and isinstance(getattr(inner, "module", None), torch.nn.Module):
        inner = inner.module
    sc = dict(kernel=bool(embed_sc and isinstance(inner, _SN)), checked=False, on=False)

    # Device-side step bookkeeping (graph replays with device-drawn noise): the captured step starts with fd_sample_advance,
    # which takes t, {g_rot(t), b(t)} and the step's draws from device arrays indexed by a device counter -- no fill_ / copy_ /
    # normal_ launch between replays; the generator fills NOISE_STEPS steps' worth of draws in one launch.
    NOISE_STEPS = 50
    adv = None

    def step_body():
        if adv is not None:
            lib.call("fd_sample_advance", adv["counter"], adv["all_t"], all_tp, adv["z_all"], NOISE_STEPS, z_both.numel(),
                     st["t"], B, tparams, z_both)
        out = model(st)      # (sc["kernel"]: the heads kernel writes the predicted CA positions into st["sc_ca_t"] itself)
        if embed_sc and sc["on"]:
            if sc["kernel"] and not sc["checked"] and not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
                sc["checked"] = True
                if not torch.equal(st["sc_ca_t"], out["rigids"][..., 4:].to(torch.float32)):
                    sc["kernel"] = False            # (the attribute was not honoured: fall back to the copy, for good)
                    inner.__dict__.pop("_fd_sc_ca_out", None)
            if not sc["kernel"]:
                st["sc_ca_t"].copy_(out["rigids"][..., 4:])
        # (in place: fd_se3_reverse_step reads every row it needs for the centring mean before it writes any)
        diffuser.reverse_device(st["rigids_t"], out["rot_score"], out["trans_score"], 0.5, dt, diffuse_mask=diffuse_mask,
                                center=center, noise_scale=noise_scale, noise=(z_rot, z_trans), tparams=tparams,
                                out=st["rigids_t"])
        return out

    try:
        traj = []
        if embed_sc and self_condition:
            set_t(steps[0])
            st["sc_ca_t"].copy_(model(st)["rigids"][..., 4:])
        if embed_sc:
            # from here on every forward leaves its predicted CA positions in st["sc_ca_t"] (an input of the NEXT forward; this
            # forward's own readers -- the edge embedder's distogram -- are launched before the heads kernel that writes it)
            st["sc_ca_t"] = st["sc_ca_t"].to(torch.float32).contiguous()
            if sc["kernel"]:
                inner._fd_sc_ca_out = st["sc_ca_t"]
        sc["on"] = True
        graph = None
        n_rev = int(np.sum(steps > min_t))
        # (the device step counter of fd_sample_advance equals the loop index only because every reverse step precedes the final one)
        assert bool(np.all(steps[:n_rev] > min_t)) and n_rev >= len(steps) - 1
        if use_graph and lib.is_device and n_rev > 3:
            # warm up on a side stream (allocator, lazily-built constant tables), then capture one step
            saved = {k: st[k].clone() for k in ("rigids_t", "sc_ca_t")}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                set_t(steps[0])
                for _ in range(2):
                    step_body()
            torch.cuda.current_stream().wait_stream(side)
            if opts.sampler_device_steps:
                adv = dict(counter=torch.zeros(1, dtype=torch.int32, device=dev),
                           all_t=torch.tensor(np.ascontiguousarray(steps), dtype=torch.float32).to(dev),
                           z_all=torch.zeros((NOISE_STEPS,) + tuple(z_both.shape), dtype=torch.float64, device=dev))
            graph = torch.cuda.CUDAGraph()
            n_launch = lib.cdll.fd_launch_count()
            with torch.cuda.graph(graph):
                cap_out = step_body()          # (static buffers: every replay rewrites them)
            if stats is not None:
                # kernels of this library in ONE captured diffusion step (network forward + reverse step + fd_sample_advance)
                stats["kernels_per_step"] = int(lib.cdll.fd_launch_count() - n_launch)
            for k, v in saved.items():
                st[k].copy_(v)
            # (capture records the launches without running them: the counter is still 0 = the index of the first step)
        out = None
        if stats is not None and lib.is_device:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        for i, t in enumerate(steps):
            if t > min_t:
                if graph is not None and adv is not None:
                    if i % NOISE_STEPS == 0:
                        if noise_fn is None:
                            adv["z_all"].normal_(generator=generator)     # the draws of the next NOISE_STEPS steps, one launch
                        else:
                            # injected draws (trajectory parity tests): the same buffer, filled in step order
                            for k in range(min(NOISE_STEPS, len(steps) - i)):
                                if steps[i + k] > min_t:
                                    zr, zt = noise_fn(i + k, (B, N, 3))
                                    adv["z_all"][k, 0].copy_(_f64(zr, dev))
                                    adv["z_all"][k, 1].copy_(_f64(zt, dev))
                else:
                    set_t(t)
                    draw(i)
                if graph is not None:
                    graph.replay()
                    out = cap_out
                else:
                    out = step_body()
                psi = out["psi"]               # (the torsion head's output of the latest forward: no copy per step)
            else:
                # reference quirk kept: the final forward still carries the previous step's t (train_se3_diffusion.py:778-779)
                out = model(st)
                st["rigids_t"].copy_(out["rigids"])
                psi = out["psi"]
            if return_traj:
                traj.append(st["rigids_t"].clone())
        if stats is not None and lib.is_device:
            ev1.record()
            ev1.synchronize()
            stats.update(loop_ms=ev0.elapsed_time(ev1), steps=len(steps), captured=graph is not None)
    finally:
        # an exception mid-trajectory must not leave the weight-derived cache, eval mode or the profiling switch behind
        model.__dict__.pop("_fd_static", None)
        inner.__dict__.pop("_fd_sc_ca_out", None)
        from . import ops as _ops
        _ops.join()                      # (options.graph_fork: nothing of an interrupted forward stays referenced on the branch)
        if was_training:
            model.train()
        lib.gemm_profile = saved_prof
    from . import train_step as ts
    atom37, _ = ts.backbone_atoms(st["rigids_t"], psi)
    res = dict(rigids=st["rigids_t"], atom37=atom37, psi=psi)
    if return_traj:
        res["rigid_traj"] = traj
    return res
