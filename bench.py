"""Benchmark of the FrameDiff hot path on MI355X (contract: see the task statement / DESIGN.md).

Workload at N GPUs (weak scaling): every rank runs the BASELINE.json configs[1] training step --
config/base.yaml full depth (4 IPA blocks), one batch of B=30 backbones of N=128 residues
(B = floor(5e5 / N^2), data/utils.py:395) -- forward + DSM loss + backward + flat-gradient RCCL
all-reduce + Adam.  value = residues/s over all ranks, inputs resident in HBM.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One JSON line on stdout (rank 0).  `roofline` = the dominant kernel (the fd_gemm tile with the largest share of the
step: the 256x128 split-bf16 MFMA kernel), timed with HIP events on its own stream in three steps right after the
timed region (launches serialised; events inside the timed region would cost ~2.5 ms per step and overlap across the two
streams); `cpu_baseline` = the oracle
(CPU port of the reference, oracle/framediff_oracle.py) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-res", type=int, default=128)
    ap.add_argument("--batch", type=int, default=0, help="0 = floor(5e5/N^2) as the reference's length_batching")
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--mode", default="train", choices=["train", "forward", "sample"])
    ap.add_argument("--num-t", type=int, default=500, help="reverse-diffusion steps per backbone (sample mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="sample mode: eager launches instead of one hipGraph per step")
    ap.add_argument("--cpu-sample-batch", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(n_res, blocks, sample_b, steps=2):
    """The oracle (CPU port) on a bounded sample: fwd + loss + bwd of `sample_b` backbones."""
    from oracle import framediff_oracle as fo
    from se3_diffusion_amd import train_step as ts
    cores = max(1, (os.cpu_count() or 2) // 2)
    torch.set_num_threads(cores)
    conf = dict(fo.CONF, num_blocks=blocks)
    P = {k: v.requires_grad_(True) for k, v in fo.synth_params(seed=0, conf=conf).items()}
    batch = ts.synthetic_batch(sample_b, n_res, "cpu", seed=0)
    gt37, _ = fo.backbone_atoms(batch["rigids_0"][..., :4], batch["rigids_0"][..., 4:],
                                batch["torsion_angles_sin_cos"][..., 2, :])

    def step():
        out = fo.score_network_forward(P, batch, conf)
        loss = ts.dsm_loss(batch, out, gt37)
        loss.backward()
        for p in P.values():
            p.grad = None

    step()
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = (time.time() - t0) / steps
    return dict(value=round(sample_b * n_res / dt, 2), unit="residues/s", cores=cores, kind="port",
                sample=f"{steps} steps of fwd+DSM loss+bwd, B={sample_b} x N={n_res}, {blocks} blocks, "
                       f"torch-CPU fp32 oracle, {cores} threads, {dt:.2f} s/step")


# fd_gemm tile code -> (kernel, dense MFMA peak in TFLOP/s of ALGORITHMIC fp32 flops).  The split kernel spends six
# bf16 MFMAs (2.5 PFLOP/s dense, MI355X_MICROARCH.md) per fp32-accurate product, so its ceiling is 2500 / 6.
_KERNELS = {
    1: ("gemm_kernel<128,128,2,2,*,*> (fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    2: ("gemm_kernel<64,64,2,2,*,*> (fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    3: ("gemm_kernel<128,32,4,1,*,*> (fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    6: ("gemm_bx3_kernel<128,*,*> (128x128 split-bf16 tile, two blocks per CU)", round(2500.0 / 6.0, 1)),
    5: ("gemm_direct_kernel<*,*> (32x32 latency tiles, fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    4: ("gemm_bx3p_kernel<*> / gemm_bx3_kernel<256,*,*,*> (256x128 tiles, fp32 operands as 3 bf16 terms, 6 x "
        "v_mfma_f32_32x32x16_bf16 per k-step; persistent blocks when a launch has >= 2 tiles per CU)",
        round(2500.0 / 6.0, 1)),
}


def dominant_gemm(prof):
    """(tile, flops, seconds, launches) of the fd_gemm tile with the largest total time + totals over all tiles."""
    by = {}
    for rec in prof:
        s = by.setdefault(rec[0], [0.0, 0.0, 0])
        s[0] += rec[3]
        s[1] += rec[4].elapsed_time(rec[5]) * 1e-3
        s[2] += 1
    tile = max(by, key=lambda k: by[k][1])
    tot_f = sum(v[0] for v in by.values())
    tot_t = sum(v[1] for v in by.values())
    return tile, by[tile][0], by[tile][1], by[tile][2], tot_f, tot_t


def bench_sample(a, rank, world, dev, lib):
    """backbones/s of the full reverse diffusion (num_t steps = num_t + 1 network forwards + num_t - 1 reverse
    steps, config/inference.yaml:18-24): every rank samples its own batch of B backbones of length N; one bench
    'step' = one complete batch."""
    from types import SimpleNamespace as ns
    from se3_diffusion_amd import sampler, train_step as ts
    from se3_diffusion_amd.data import se3_diffuser
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    N = a.n_res
    B = a.batch if a.batch > 0 else 1
    dconf = ns(diffuse_trans=True, diffuse_rot=True, r3=ns(min_b=0.1, max_b=20.0, coordinate_scaling=0.1),
               so3=ns(num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5, schedule="logarithmic",
                      cache_dir=os.environ.get("FD_IGSO3_CACHE", "/tmp/fd_igso3_cache_bench"), use_cached_score=False))
    t_tab = time.perf_counter()
    diff = se3_diffuser.SE3Diffuser(dconf)
    t_tab = time.perf_counter() - t_tab
    torch.manual_seed(0)
    model = ScoreNetwork(ts.base_model_conf(a.blocks), diff).to(dev)
    ts.perturb_final_layers(model, seed=0)
    model.eval()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)

    def step():
        feats = sampler.init_feats(diff, B, N, dev, generator=gen)
        return sampler.sample(model, diff, feats, num_t=a.num_t, min_t=0.01, noise_scale=0.1, generator=gen,
                              use_graph=not a.no_graph)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    # per-kernel timing of the dominant kernel: HIP events around every fd_gemm launch of three network forwards
    # on the stream they run on (the timed region below replays captured graphs, where events cannot be read back)
    pf = sampler.init_feats(diff, B, N, dev, generator=gen)
    lib.gemm_profile = []
    with torch.no_grad():
        for _ in range(3):
            model(pf)
    torch.cuda.synchronize()
    prof, lib.gemm_profile = lib.gemm_profile, None
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out["rigids"]).all()
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank != 0:
        return
    tile, use_f, use_t, _n, tot_f, tot_t = dominant_gemm(prof)
    kname, peak = _KERNELS[tile]
    achieved = use_f / max(use_t, 1e-9) / 1e12
    res = {
        "metric": f"backbones/sec {a.num_t}-step sampling @ N={N}", "value": round(world * B * a.steps / dt, 4),
        "unit": "backbones/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"reverse diffusion, {a.num_t} steps (min_t 0.01, noise_scale 0.1, self-conditioning), per-GPU "
                               f"batch of {B} backbones x N={N}, config/base.yaml ScoreNetwork ({a.blocks} blocks), device-resident loop",
                   "parallelism": f"replicas x{world}", "ms_per_diffusion_step": round(dt / a.steps / a.num_t * 1e3, 3),
                   "igso3_table_build_s": round(t_tab, 2),
                   "arithmetic": "fp32 storage and accumulation everywhere; pair-level GEMMs = 3-term bf16 split on the bf16 MFMA (fp32-accurate, FD_GEMM_EXACT_F32=1 forces the fp32 MFMA), all other GEMMs fp32 MFMA, IGSO(3) fp64"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None, "kernel": kname,
                     "measured_on": "3 eager network forwards (HIP events per fd_gemm launch)",
                     "gemm_time_frac_of_step": round(tot_t / 3 * (a.num_t + 1) * a.steps / dt, 4),
                     "step_model_tflops": round(tot_f / 3 * (a.num_t + 1) * a.steps / dt / 1e12, 2)},
    }
    print(json.dumps(res), flush=True)


def main():
    a = parse()
    from se3_diffusion_amd import dist as fdist
    rank, world, local = fdist.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs an AMD GPU (the hot path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from se3_diffusion_amd import hip, loss as floss, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    lib = hip.get_lib()

    if a.mode == "sample":
        return bench_sample(a, rank, world, dev, lib)
    N = a.n_res
    B = a.batch if a.batch > 0 else max(1, 500000 // (N * N))
    torch.manual_seed(0)
    model = ScoreNetwork(ts.base_model_conf(a.blocks), diffuser=None).to(dev)
    ts.perturb_final_layers(model, seed=0)
    fdist.broadcast_params(model)
    model.train()
    batch = ts.synthetic_batch(B, N, dev, seed=100 + rank)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])
    from se3_diffusion_amd.optim import FlatAdam
    # Adam (torch.optim.Adam's rule, lr 1e-4 as train_se3_diffusion.py:139) over flat parameter / gradient / moment
    # buffers: param.grad are views of ONE buffer (single RCCL all-reduce), the update is one launch
    opt = FlatAdam(model.parameters(), lr=1e-4)
    grads = opt
    model.accumulate_into_grad = True      # backward kernels accumulate straight into the flat all-reduce buffer

    def step():
        if a.mode == "forward":
            with torch.no_grad():
                model(batch)
            return
        grads.zero()
        out = model(batch)
        loss = floss.dsm_loss(batch, out, gt37)     # fused Experiment.loss_fn arithmetic (fd_dsm_loss)
        loss.backward()
        grads.all_reduce_mean()
        opt.step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    # Per-launch HIP events are NOT recorded inside the timed region: 1200 event records per step cost ~2.5 ms of it
    # (measured: 42.1 vs 39.3 ms in exact-fp32 mode), and with the weight-gradient GEMMs on a second stream a launch's
    # start/stop events would span the kernels it shares the GPU with.  The roofline of the dominant kernel is taken
    # from three more steps right after it, same shapes and data, with that side stream switched off (launches
    # serialised) and events around every fd_gemm launch on the stream it runs on.
    from se3_diffusion_amd import ops as fops
    side_was = fops.set_grad_stream(False)
    lib.gemm_profile = []
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof, lib.gemm_profile = lib.gemm_profile, None
    fops.set_grad_stream(side_was)
    # the same step with every GEMM on the exact fp32 MFMA (bitwise fmaf-chain products: FD_GEMM_EXACT_F32=1), so the
    # line carries both arithmetic choices; not part of `value`
    exact_ms = None
    if not os.environ.get("FD_BENCH_PROFILE"):     # (rocprofv3 runs: keep the kernel trace to the shipped arithmetic)
        was_exact = lib.cdll.fd_gemm_set_exact_f32(1)
        step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        barrier()
        exact_ms = (time.perf_counter() - t0) / 3 * 1e3
        lib.cdll.fd_gemm_set_exact_f32(was_exact)

    if rank != 0:
        return
    if os.environ.get("FD_BENCH_GEMM_DETAIL"):
        shapes = {}
        for p in prof:
            s = shapes.setdefault((p[0], p[1], p[2]) + p[6], [0.0, 0.0, 0])
            s[0] += p[3]; s[1] += p[4].elapsed_time(p[5]) * 1e-3; s[2] += 1
        rows = sorted(shapes.items(), key=lambda kv: -kv[1][1])
        sys.stderr.write("tile akc bkc (M,N,K,batch,gate,beta,pair,ksplit)  calls/step  ms/step  TF/s\n")
        for k, v in rows[:40]:
            sys.stderr.write(f"{k}  {v[2] / 3:.1f}  {v[1] / 3 * 1e3:.3f}  {v[0] / v[1] / 1e12:.1f}\n")
    tile, dflops, dtime, dn, all_flops, tot_t = dominant_gemm(prof)
    kname, peak = _KERNELS[tile]
    achieved = dflops / dtime / 1e12
    nprof = 3
    ms = dt / a.steps * 1e3
    res = {
        "metric": "residues/sec IPA fwd+bwd" if a.mode == "train" else "residues/sec IPA fwd",
        "value": round(world * B * N * a.steps / dt, 1), "unit": "residues/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config/base.yaml ScoreNetwork ({a.blocks} IPA blocks, 17.4M params), per-GPU batch "
                               f"B={B} x N={N} residues, {'fwd + fused DSM loss + bwd + RCCL grad all-reduce + Adam' if a.mode == 'train' else 'forward only'}",
                   "parallelism": f"dp{world}", "global_batch": world * B, "n_res": N,
                   "ms_per_step_exact_f32_gemms": None if exact_ms is None else round(exact_ms, 3),
                   "arithmetic": "fp32 storage and accumulation everywhere; pair-level GEMMs = 3-term bf16 split on the bf16 MFMA (fp32-accurate, FD_GEMM_EXACT_F32=1 forces the fp32 MFMA), all other GEMMs fp32 MFMA, IGSO(3) fp64"},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None, "kernel": kname,
                     "vs_fp32_mfma_peak": round(achieved / 157.3, 4),
                     "traffic_note": "PMC (offline, profiles/r01_pmc_fd_gemm.md): 1.62 GB HBM per M=491520,N=K=384 launch "
                                     "vs 1.51 GB algorithmic (FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                     "measured_on": "3 steps right after the timed region with the gradient side stream off "
                                    "(HIP events per fd_gemm launch on its stream)",
                     "launches_per_step": dn // nprof,
                     "avg_launch_us": round(dtime / max(1, dn) * 1e6, 2),
                     "algorithmic_flops_per_launch": round(dflops / max(1, dn), 1),
                     "serialised_gemm_ms_per_step": round(tot_t / nprof * 1e3, 3),
                     "all_gemm_tflops": round(all_flops / max(tot_t, 1e-9) / 1e12, 2),
                     "step_model_tflops": round(all_flops / nprof / (dt / a.steps) / 1e12, 2)},
    }
    if not a.no_cpu_baseline and world == 1:
        try:
            res["cpu_baseline"] = cpu_baseline(N, a.blocks, a.cpu_sample_batch)
        except Exception as e:  # noqa: BLE001 -- the baseline must not lose the GPU measurement
            res["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
