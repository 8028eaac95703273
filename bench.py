"""Benchmark of the FrameDiff hot path on MI355X (contract: see the task statement / DESIGN.md).

Workload at N GPUs (weak scaling): every rank runs the BASELINE.json configs[1] training step --
config/base.yaml full depth (4 IPA blocks), one batch of B=30 backbones of N=128 residues
(B = floor(5e5 / N^2), data/utils.py:395) -- forward + DSM loss + backward + flat-gradient RCCL
all-reduce + Adam.  value = residues/s over all ranks, inputs resident in HBM.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --mode sample --n-res 256 --batch 1 --steps 1 --warmup 1        # full 500-step sampling
  python bench.py --mixed-n                                                         # BASELINE configs[3]

Before the W warm-up steps the process runs FD_BENCH_PRIME (default 4) untimed priming steps (code-object loads, allocator
growth) and then chunks of 4 more until two consecutive chunks agree within 1.5 % (GPU power state back from idle: at least 1.5 s, at most 4 s --
FD_BENCH_PRIME_MIN_S / FD_BENCH_PRIME_MAX_S); the timed region is exactly K steps between two barriers.

One JSON line on stdout (rank 0):
  * value / ms_per_step: the training step (the first half of BASELINE.json's metric);
  * config.sampling: the second half -- backbones/s of 500-step reverse diffusion at N = 128 / 256 / 512, from bounded
    device-resident runs (--sample-steps diffusion steps each, scaled to 501 network forwards; rank 0 at N = 1 GPU only);
  * config.self_conditioning_50pct: the step with the reference's 50 % extra no-grad forward (train_se3_diffusion.py:535-537);
  * roofline: the dominant kernel (largest share of the serialised step among fd_gemm's tiles and the fused
    edge-transition kernel), timed with HIP events on its own stream in three steps right after the timed region;
  * cpu_baseline: the unmodified reference when FD_REFERENCE_ROOT (default /root/reference) exists, else the oracle (CPU port),
    on a bounded sample of the same workload, best of a thread sweep.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ARITH = ("fp32 storage and accumulation everywhere; pair-level GEMMs and the fused edge transition = 3-term bf16 split on the "
         "bf16 MFMA (fp32-accurate; FD_GEMM_EXACT_F32=1 forces bitwise-fp32 MFMA chains), all other GEMMs fp32 MFMA, IGSO(3) fp64")


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
    ap.add_argument("--no-sampling", action="store_true", help="skip the bounded sampling runs of the default line")
    ap.add_argument("--sample-steps", type=int, default=50, help="diffusion steps of each bounded sampling run")
    ap.add_argument("--no-graph", action="store_true", help="sample mode: eager launches instead of one hipGraph per step")
    ap.add_argument("--cpu-sample-batch", type=int, default=4)
    ap.add_argument("--mixed-n", action="store_true",
                    help="BASELINE configs[3]: every step all ranks draw the same N in [100, 512], B = min(32, 5e5 // N^2)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_res, blocks, sample_b, budget_s=40.0, reps=3, force_kind=None):
    """fwd + DSM loss + bwd of `sample_b` backbones on the host: the unmodified reference if it is on this machine
    (FD_REFERENCE_ROOT), else the oracle (CPU port); best of a thread sweep, bounded by `budget_s` seconds."""
    from oracle import framediff_oracle as fo
    from oracle import ref_loader as rl
    from se3_diffusion_amd import train_step as ts
    ncpu = os.cpu_count() or 2
    conf = dict(fo.CONF, num_blocks=blocks)
    batch = ts.synthetic_batch(sample_b, n_res, "cpu", seed=0)
    gt37, _ = fo.backbone_atoms(batch["rigids_0"][..., :4], batch["rigids_0"][..., 4:],
                                batch["torsion_angles_sin_cos"][..., 2, :])
    kind = "port"
    step = None
    if rl.available() and force_kind != "port":
        try:
            rl.install()
            from data import se3_diffuser as ref_se3      # the reference's own modules
            from model import score_network as ref_sn
            rconf = rl.base_conf(os.environ.get("FD_IGSO3_CACHE", "/tmp/fd_igso3_cache_bench"), num_blocks=blocks)
            ref_model = ref_sn.ScoreNetwork(rconf.model, ref_se3.SE3Diffuser(rconf.diffuser))
            ref_model.load_state_dict(fo.synth_params(seed=0, conf=conf), strict=True)
            ref_model.train()

            def step():
                ref_model.zero_grad(set_to_none=True)
                loss = ts.dsm_loss(batch, ref_model(batch), gt37)
                loss.backward()
            kind = "reference"
        except Exception:  # noqa: BLE001 -- fall back to the port
            step = None
    if step is None:
        P = {k: v.requires_grad_(True) for k, v in fo.synth_params(seed=0, conf=conf).items()}

        def step():
            out = fo.score_network_forward(P, batch, conf)
            loss = ts.dsm_loss(batch, out, gt37)
            loss.backward()
            for p in P.values():
                p.grad = None

    # "all host cores", median of several steps.  The sweep goes from the settings torch's CPU kernels peak at
    # on a 2-socket EPYC box (16-32 threads at these sizes) up to every logical CPU (ncpu); each setting is the MEDIAN of up to
    # `reps` steps; a setting is abandoned after its first step when that step alone is > 1.5 x the best median so far (all 256
    # logical CPUs: 230 s per step in round 3, oversubscribed oneDNN / OpenMP teams), and the sweep ends when the budget is spent.
    sweep = [t for t in (32, 16, 64, 128, ncpu) if t <= ncpu] or [ncpu]
    sweep = list(dict.fromkeys(sweep))
    t_start = time.time()
    best, tried = None, []
    for i, th in enumerate(sweep):
        torch.set_num_threads(th)
        if i == 0:
            step()                               # warm-up (allocator, oneDNN primitives)
        dts = []
        for r in range(reps):
            t0 = time.time()
            step()
            dts.append(time.time() - t0)
            spent = time.time() - t_start
            if (best is not None and dts[0] > 1.5 * best[1]) or spent + dts[-1] > budget_s:
                break
        med = sorted(dts)[len(dts) // 2]
        tried.append((th, round(med, 2), len(dts)))
        if best is None or med < best[1]:
            best = (th, med)
        # the sweep ends when the budget is spent or when more threads are clearly slower: past the peak every doubling is slower
        # still, and ALL logical CPUs cost minutes per step on the 256-CPU GPU box (measured once per round and written down
        # instead of re-measured by every default run: 230 s per step in round 3, 228.7 s in round 4, profiles/r04_bench_train.json)
        if time.time() - t_start + med > budget_s or med > 1.5 * best[1]:
            break
    th, dt = best
    skipped = [t for t in sweep if t not in [x[0] for x in tried]]
    return dict(value=round(sample_b * n_res / dt, 2), unit="residues/s", cores=th, kind=kind,
                logical_cpus=ncpu, thread_settings_not_reached=skipped,
                all_logical_cpus_note="every logical CPU of the 2 x EPYC 9575F box (256 threads) measured 228.7 s per step in round 4 "
                                      "(profiles/r04_bench_train.json) against 1.5 s at 16 threads: the sweep stops once more threads "
                                      "are > 1.5 x slower than the best",
                port_vs_reference=_port_vs_reference(),
                sample=f"fwd + DSM loss + bwd, B={sample_b} x N={n_res}, {blocks} blocks, "
                       f"{'unmodified reference' if kind == 'reference' else 'torch-CPU fp32 oracle (port)'}; best median over threads "
                       f"{tried} (threads, median s/step, steps timed) on {ncpu} logical CPUs ({_cpu_model()}), budget {budget_s:.0f} s")


def _port_vs_reference():
    """How the CPU port (oracle) compares with the UNMODIFIED reference where both can run (the build container has the reference
    checkout, the GPU box does not): tools/cpu_port_vs_reference.py times both on the same cores and commits the result under
    profiles/; the bench line carries it so that `kind: "port"` is self-documenting."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_port_vs_reference.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    d["source"] = os.path.relpath(files[-1], ROOT)
    return d


# ---------------------------------------------------------------------------------------------------------------------
# roofline bookkeeping
# ---------------------------------------------------------------------------------------------------------------------
# profile tile code -> (kernel, dense MFMA peak in TFLOP/s of ALGORITHMIC fp32 flops).  The split-bf16 kernels spend six
# bf16 MFMAs (2.5 PFLOP/s dense, MI355X_MICROARCH.md) per fp32-accurate product, so their ceiling is 2500 / 6.
_KERNELS = {
    1: ("gemm_kernel<128,128,2,2,*,*> (fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    2: ("gemm_kernel<64,64,2,2,*,*> (fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    3: ("gemm_kernel<128,32,4,1,*,*> (fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    6: ("gemm_bx3_kernel<128,*,*> (128x128 split-bf16 tile, two blocks per CU)", round(2500.0 / 6.0, 1)),
    5: ("gemm_direct_kernel<*,*> (32x32 latency tiles, fp32 v_mfma_f32_32x32x2_f32)", 157.3),
    4: ("gemm_bx3p_kernel<*> / gemm_bx3_kernel<256,*,*,*> (256x128 tiles, fp32 operands as 3 bf16 terms, 6 x "
        "v_mfma_f32_32x32x16_bf16 per k-step; persistent blocks when a launch has >= 2 tiles per CU)",
        round(2500.0 / 6.0, 1)),
    7: ("edge_mlp16_kernel<*> (fused edge transition 128->384->384->128 + LayerNorm per pair row, register-chained, "
        "fp32 operands as 3 bf16 terms, 6 x v_mfma_f32_16x16x32_bf16 per k-step, weights streamed by LDS-DMA)",
        round(2500.0 / 6.0, 1)),
    8: ("edge_embed_kernel (fused edge embedder: relpos + distogram features generated in registers, 3 register-chained "
        "layers + LayerNorm, split-bf16 v_mfma_f32_16x16x32_bf16)", round(2500.0 / 6.0, 1)),
    10: ("gemm_s64_kernel<*,*> (64x64 split-bf16 tile, node-level / attention GEMMs, 6 x v_mfma_f32_32x32x16_bf16 per "
         "k-step)", round(2500.0 / 6.0, 1)),
    11: ("group_dw_kernel (grouped node-level weight gradients: 128x128 tiles over a (tile, stage) sequence shared in equal pieces, "
         "split-bf16 v_mfma_f32_32x32x16_bf16)", round(2500.0 / 6.0, 1)),
    9: ("pair_dw_kernel (grouped pair-row weight gradients: 384x128 tiles, float4 staging + ds_read_b64_tr_b16 operands, "
        "split-bf16 v_mfma_f32_32x32x16_bf16)", round(2500.0 / 6.0, 1)),
}
# Sustained v_mfma_f32_32x32x16_bf16 rate of this chip on random operands (tools/probes/mfma_peak.hip, profiles/: 1.76 PFLOP/s,
# power-limited; 2.22 on zeros): the practical ceiling of the split-bf16 kernels is 1760 / 6 TFLOP/s of algorithmic fp32 flops
SUSTAINED_SPLIT_PEAK = round(1760.0 / 6.0, 1)
# profile tile code -> kernel-name prefix in rocprofv3's tables (tools/pmc_roofline.py keys its JSON by them)
_PMC_NAMES = {7: ("edge_mlp16_kernel<false", "edge_mlp16_kernel<true"), 9: ("pair_dw_kernel",), 11: ("group_dw_kernel",),
              8: ("edge_embed_kernel",), 4: ("gemm_bx3p_kernel", "gemm_bx3_kernel<256"), 6: ("gemm_bx3_kernel<128",),
              10: ("gemm_s64_kernel",), 2: ("gemm_kernel<64, 64",), 1: ("gemm_kernel<128, 128",), 3: ("gemm_kernel<128, 32",),
              5: ("gemm_direct_kernel",)}
# algorithmic HBM bytes per pair row of the fused edge-transition launches in TRAINING: forward reads z
# (512 B) and writes z' (512), the saves h1, h2 (2 x 1536), y (512), mean / rstd (8), the packed signs of h1 / h2 (96) and the
# next block's zb (160); backward reads dy (512) and the packed gates (96), writes d2, d1 (3072) and dz (512); backward with the
# fused LayerNorm-backward / dzb W40 prologue (template arguments 4, 5) reads the upstream gradient (512), dzb (160), y (512),
# mean / rstd (8) and the gates (96), writes dy (512), d2, d1 (3072) and dz (512)
_EDGE_ALGO_BYTES = {"edge_mlp16_kernel<false": 512 + 512 + 3072 + 512 + 8 + 96 + 160,
                    # (template arguments: BWD, ZB, TRAIN, LNB, ZBW)
                    "edge_mlp16_kernel<true, false, true, true, true": 512 + 160 + 512 + 8 + 96 + 512 + 3072 + 512,
                    "edge_mlp16_kernel<true, false, true, true, false": 512 + 512 + 8 + 96 + 512 + 3072 + 512,
                    "edge_mlp16_kernel<true, false, true, false": 512 + 96 + 3072 + 512}


def pmc_traffic(tile, rows):
    """HBM bytes per launch of the kernel class `tile` from the newest committed PMC table (profiles/r*_pmc_traffic.json, made
    by tools/pmc_roofline.sh on the training step at B=30 x N=128: FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on
    known 1 GiB transfers).  Launch-weighted mean over the variants of the class that the step runs (forward with saves +
    backward for the fused edge kernels) -- the same population `achieved` averages over."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files or tile not in _PMC_NAMES:
        return None
    with open(files[-1]) as f:
        tab = json.load(f)
    if rows != 30 * 128 * 128 and tile in (7, 8, 9):
        return None                               # the table was taken at B=30 x N=128
    var, nl, tb, raw = {}, 0.0, 0.0, 0.0
    for name, k in tab["kernels"].items():
        if any(name.startswith(pfx) for pfx in _PMC_NAMES[tile]) and k["launches_per_step"] >= 0.9:
            # (< 0.9 launches per step: a variant that only the occasional extra no-grad forward of the traced run used)
            var[name] = {"launches_per_step": k["launches_per_step"], "bytes_per_launch": round(k["bytes_per_launch"]),
                         "raw_counter_bytes_per_launch": round(k["raw_bytes_per_launch"])}
            for pfx, nbytes in _EDGE_ALGO_BYTES.items():
                if name.startswith(pfx):
                    var[name]["algorithmic_bytes_per_launch"] = nbytes * rows
            nl += k["launches_per_step"]
            tb += k["fetch_per_step"] + k["write_per_step"]
            raw += k["fetch_raw_per_step"] + k["write_raw_per_step"]
    if nl == 0:
        return None
    cal = tab.get("calibration", {})
    return {"bytes_per_launch": round(tb / nl), "raw_counter_bytes_per_launch": round(raw / nl), "by_variant": var,
            "source": os.path.relpath(files[-1], ROOT),
            "calibration_factors": {c: {p_: round(v.get("factor") or 0.0, 3) for p_, v in cal.get(c, {}).items()} for c in cal}}


def dominant_kernel(prof):
    """(tile, flops, seconds, launches) of the profiled kernel class with the largest total time + totals over all."""
    by = {}
    for rec in prof:
        s = by.setdefault(rec[0], [0.0, 0.0, 0, rec[6]])
        s[0] += rec[3]
        s[1] += rec[4].elapsed_time(rec[5]) * 1e-3
        s[2] += 1
    tile = max(by, key=lambda k: by[k][1])
    tot_f = sum(v[0] for v in by.values())
    tot_t = sum(v[1] for v in by.values())
    return tile, by[tile][0], by[tile][1], by[tile][2], tot_f, tot_t, by[tile][3], {k: round(v[1] * 1e3, 3) for k, v in by.items()}


def make_diffuser():
    from types import SimpleNamespace as ns
    from se3_diffusion_amd.data import se3_diffuser
    dconf = ns(diffuse_trans=True, diffuse_rot=True, r3=ns(min_b=0.1, max_b=20.0, coordinate_scaling=0.1),
               so3=ns(num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5, schedule="logarithmic",
                      cache_dir=os.environ.get("FD_IGSO3_CACHE", "/tmp/fd_igso3_cache_bench"), use_cached_score=False))
    t0 = time.perf_counter()
    diff = se3_diffuser.SE3Diffuser(dconf)
    return diff, time.perf_counter() - t0


# fp32-MFMA floors of one network forward (algorithmic fwd GFLOP per backbone / 157.3 TFLOP/s), ms
_FWD_FLOOR_MS = {128: 0.26, 256: 0.99, 512: 3.87}


def sampling_rates(dev, blocks, num_t_run, lib, cases=((128, 1, True), (128, 8, False), (128, 32, False), (256, 1, True),
                                                        (512, 1, False), (512, 8, False))):
    """The sampling half of the metric.  Cases flagged True run the FULL 500-step trajectory (config/inference.yaml:18-24:
    N=128 and N=256 at B=1, under two seconds together); the others run num_t_run diffusion steps (num_t_run + 1 network
    forwards) and are scaled to the 501 forwards of a 500-step trajectory.  Every case carries the roofline of the dominant
    kernel of its network forward (HIP events around every profiled launch of three eager forwards, as the training line)."""
    from se3_diffusion_amd import sampler, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    diff, _ = make_diffuser()
    torch.manual_seed(0)
    model = ScoreNetwork(ts.base_model_conf(blocks), diff).to(dev)
    ts.perturb_final_layers(model, seed=0)
    model.eval()
    gen = torch.Generator(device=dev).manual_seed(99)
    out = {}
    for N, B, full in cases:
        steps_run = 500 if full else num_t_run

        def run(num_t, st=None):
            feats = sampler.init_feats(diff, B, N, dev, generator=gen)
            return sampler.sample(model, diff, feats, num_t=num_t, min_t=0.01, noise_scale=0.1, generator=gen, use_graph=True,
                                  stats=st)
        run(min(steps_run, 20))                  # warm-up: allocator, tables, code objects
        st = {}
        r = run(steps_run, st)
        torch.cuda.synchronize()
        assert torch.isfinite(r["rigids"]).all()
        # loop_ms = the reverse loop (steps_run network forwards incl. the last, frames-only one); the self-conditioning
        # warm-up forward and the one-off graph capture of a trajectory are outside it -> 501 forwards per trajectory
        per_fwd = st["loop_ms"] * 1e-3 / steps_run
        pf = sampler.init_feats(diff, B, N, dev, generator=gen)
        lib.gemm_profile = []
        with torch.no_grad():
            for _ in range(3):
                model(pf)
        torch.cuda.synchronize()
        prof, lib.gemm_profile = lib.gemm_profile, None
        tile, use_f, use_t, n_l, tot_f, tot_t, _shape, _by = dominant_kernel(prof)
        kname, peak = _KERNELS[tile]
        ach = use_f / max(use_t, 1e-9) / 1e12
        floor = _FWD_FLOOR_MS[N] * B
        out[f"N{N}_B{B}"] = {
            "backbones_per_s": round(B / (per_fwd * 501), 4), "ms_per_diffusion_step": round(per_fwd * 1e3, 3),
            "measured_steps": steps_run, "full_trajectory": bool(full),
            # kernels of the library in one captured diffusion step = network forward + fd_sample_advance + fd_se3_reverse_step
            # (fd_launch_count across the hipGraph capture, sampler.sample)
            "kernels_per_step": st.get("kernels_per_step"), "kernels_per_forward": (st["kernels_per_step"] - 2) if st.get("kernels_per_step") else None,
            "frac_of_fp32_mfma_floor": round(floor / (per_fwd * 1e3), 4),
            "roofline": {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "launches_per_forward": n_l // 3,
                         "avg_launch_us": round(use_t / max(1, n_l) * 1e6, 2),
                         "share_of_profiled_gemm_time": round(use_t / max(tot_t, 1e-9), 3),
                         "forward_model_tflops": round(tot_f / 3 / per_fwd / 1e12, 2),
                         "measured_on": "3 eager forwards (HIP events per profiled launch); the timed loop replays hipGraphs"}}
    return out


def cpu_sampling_baseline(n_res, blocks, steps=10, budget_s=40.0):
    """Reverse diffusion on the host: `steps` reverse steps of one N-residue backbone -- the unmodified reference's
    Experiment.inference_fn loop body when /root/reference is on this machine (model forward + SE3Diffuser.reverse), else
    the oracle's forward (CPU port of ScoreNetwork) + this package's numpy diffuser entry points (the reference's own host
    path for CPU-resident frames) -- scaled to the 501 forwards + 499 reverse steps of a 500-step trajectory."""
    import numpy as np
    from oracle import framediff_oracle as fo
    from oracle import ref_loader as rl
    from se3_diffusion_amd.openfold.utils import rigid_utils as ru
    ncpu = os.cpu_count() or 2
    conf = dict(fo.CONF, num_blocks=blocks)
    P = fo.synth_params(seed=0, conf=conf)
    kind, fwd, diff = "port", None, None
    if rl.available():
        try:
            rl.install()
            from data import se3_diffuser as ref_se3
            from model import score_network as ref_sn
            rconf = rl.base_conf(os.environ.get("FD_IGSO3_CACHE", "/tmp/fd_igso3_cache_bench"), num_blocks=blocks)
            diff = ref_se3.SE3Diffuser(rconf.diffuser)
            ref_model = ref_sn.ScoreNetwork(rconf.model, diff)
            ref_model.load_state_dict(P, strict=True)
            ref_model.eval()
            fwd = lambda f: ref_model(f)
            from openfold.utils import rigid_utils as ru  # noqa: F811 -- the reference's own Rigid
            kind = "reference"
        except Exception:  # noqa: BLE001
            fwd = None
    if fwd is None:
        diff, _ = make_diffuser()
        fwd = lambda f: fo.score_network_forward(P, f, conf, tfmr_mask_mode="bool")
    np.random.seed(0)
    N = n_res
    feats = dict(res_mask=torch.ones(1, N), fixed_mask=torch.zeros(1, N), seq_idx=torch.arange(1, N + 1)[None],
                 torsion_angles_sin_cos=torch.zeros(1, N, 7, 2), sc_ca_t=torch.zeros(1, N, 3),
                 rigids_t=diff.sample_ref(n_samples=N, as_tensor_7=True)["rigids_t"].reshape(1, N, 7).float(), t=torch.ones(1))
    ts_ = np.linspace(0.01, 1.0, 500)[::-1]

    def one(i):
        feats["t"] = float(ts_[i]) * torch.ones(1)
        with torch.no_grad():
            o = fwd(feats)
        feats["sc_ca_t"] = o["rigids"][..., 4:]
        rg = diff.reverse(rigid_t=ru.Rigid.from_tensor_7(feats["rigids_t"]), rot_score=o["rot_score"].numpy(),
                          trans_score=o["trans_score"].numpy(), diffuse_mask=np.ones((1, N)), t=float(ts_[i]), dt=1 / 500,
                          center=True, noise_scale=0.1)
        feats["rigids_t"] = rg.to_tensor_7().float()

    sweep = [t for t in (16, 32, 64, 8) if t <= ncpu] or [ncpu]
    t_start, best, tried = time.time(), None, []
    for k, th in enumerate(sweep):
        torch.set_num_threads(th)
        if k == 0:
            one(0)
        t0 = time.time()
        n = steps if k == 0 else max(3, steps // 3)
        for i in range(n):
            one(1 + i)
        dt = (time.time() - t0) / n
        tried.append((th, round(dt, 4)))
        if best is None or dt < best[1]:
            best = (th, dt)
        if time.time() - t_start > budget_s:
            break
    th, dt = best
    return dict(value=round(1.0 / (dt * 501), 5), unit="backbones/s", cores=th, kind=kind,
                sample=f"{steps} reverse-diffusion steps (model forward + SE3Diffuser.reverse) of one N={n_res} backbone, {blocks} "
                       f"blocks, {'unmodified reference' if kind == 'reference' else 'oracle forward (CPU port) + numpy diffuser'}, "
                       f"scaled to 501 forwards; best of threads {tried} (threads, s/step) on {ncpu} logical CPUs ({_cpu_model()})")


def bench_sample(a, rank, world, dev, lib):
    """backbones/s of the full reverse diffusion (num_t steps = num_t + 1 network forwards + num_t - 1 reverse
    steps, config/inference.yaml:18-24): every rank samples its own batch of B backbones of length N; one bench
    'step' = one complete batch."""
    from se3_diffusion_amd import sampler, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    N = a.n_res
    B = a.batch if a.batch > 0 else 1
    diff, t_tab = make_diffuser()
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
    # per-kernel timing of the dominant kernel: HIP events around every profiled launch of three network forwards
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
    tile, use_f, use_t, _n, tot_f, tot_t, _shape, _by = dominant_kernel(prof)
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
                   "igso3_table_build_s": round(t_tab, 2), "arithmetic": ARITH},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None, "kernel": kname,
                     "measured_on": "3 eager network forwards (HIP events per profiled launch)",
                     "step_model_tflops": round(tot_f / 3 * (a.num_t + 1) * a.steps / dt / 1e12, 2)},
    }
    print(json.dumps(res), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start the N ranks here -- re-run this command line under
    torch.distributed.run (one process per GPU, LOCAL_RANK -> device, backend nccl = RCCL; train_se3_diffusion.py:83-91,
    273-277 is the reference's DDP launch this replaces) and hand its exit code back.  Rank 0's JSON line passes through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's device-buffer sharing between the ranks
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    from se3_diffusion_amd import dist as fdist
    rank, world, local = fdist.init_from_env()
    if world != a.gpus:
        # the launcher decides (torchrun --nproc-per-node=N bench.py without --gpus N is a valid way to start N ranks)
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks; measuring {world} ranks\n")
        a.gpus = world
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
    from se3_diffusion_amd.optim import FlatAdam
    # Adam (torch.optim.Adam's rule, lr 1e-4 as train_se3_diffusion.py:139) over flat parameter / gradient / moment
    # buffers: param.grad are views of ONE buffer (single RCCL all-reduce), the update is one launch
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=model.flat_layout_groups())
    grads = opt
    model.accumulate_into_grad = True      # backward kernels accumulate straight into the flat all-reduce buffer
    overlap = None
    if world > 1 and os.environ.get("FD_DP_OVERLAP", "0") != "0":
        # opt-in: the all-reduce of a parameter group starts as soon as the backward pass has issued its last gradient launch.
        # Off by default: one 69.8 MB all-reduce after the backward is <= 0.8 ms of a ~25 ms step over xGMI, and the
        # overlapped form has only ever run over gloo (tests/test_dist.py, two processes on one GPU), never over RCCL
        overlap = fdist.OverlapAllReduce(model, opt)
        model._fd_grad_ready = overlap.ready

    def make_batch(n, b, seed):
        bt = ts.synthetic_batch(b, n, dev, seed=seed)
        g37, _ = ts.backbone_atoms(bt["rigids_0"], bt["torsion_angles_sin_cos"][..., 2, :])
        return bt, g37

    if a.mixed_n:
        # BASELINE configs[3]: the reference's DDP sampler hands every rank the same protein in a step
        # (pdb_data_loader.py:467,483), i.e. all ranks share N; B = min(batch_size = 32, max_squared_res // N^2)
        # (data/utils.py:395, base.yaml:83-84).  Lengths come from one seeded stream shared by all ranks; the batches are
        # pre-generated so the timed region holds only the step.
        sched = fdist.mixed_length_schedule(a.warmup + a.steps)
        data = [make_batch(n, b, 1000 * i + rank) for i, (n, b) in enumerate(sched)]
    else:
        sched = [(N, B)] * (a.warmup + a.steps)
        data = [make_batch(N, B, 100 + rank)] * (a.warmup + a.steps)

    def step(i=0, self_cond=False):
        batch, gt37 = data[i % len(data)]
        if a.mode == "forward":
            with torch.no_grad():
                model(batch)
            return
        if self_cond:                                # train_se3_diffusion.py:519-522,535-537
            with torch.no_grad():
                sc = model(batch)["rigids"][..., 4:]
            batch = dict(batch, sc_ca_t=sc)
        grads.zero()
        out = model(batch)
        loss = floss.dsm_loss(batch, out, gt37)     # fused Experiment.loss_fn arithmetic (fd_dsm_loss)
        loss.backward()
        if overlap is not None:
            overlap.finish()
        else:
            grads.all_reduce_mean()
        opt.step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # FD_BENCH_MAIN_PRIORITY=n: run the step on a stream of that HIP priority instead of the default stream (A/B of stream
    # priorities between the dX chain and the weight-gradient side stream)
    main_prio = os.environ.get("FD_BENCH_MAIN_PRIORITY")
    if main_prio is not None:
        hp = torch.cuda.Stream(device=dev, priority=int(main_prio))
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    # Priming (part of set-up, before the W warm-up steps the contract names): the first steps of a process load ~60 code
    # objects, grow the caching allocator to its steady state (~40 GB at B=30 x N=128) and bind the flat optimiser's
    # views; with W = 3 one of those could still land in the timed region (one 28.7 ms outlier against 25.9 ms).
    n_prime = int(os.environ.get("FD_BENCH_PRIME", "4"))
    for i in range(n_prime):
        step(i)
    barrier()
    # ... and the GPU's clocks: after ~20 s of idle time (the interpreter start-up of this very process behind another job) the
    # first ~0.5 s of steps run 12 % slow (25.7 against 22.9 ms per step measured behind a 20 s sleep; profiles/r04_ab.txt).
    # Priming continues in chunks of 4 steps for at least FD_BENCH_PRIME_MIN_S (1.5 s) and until two consecutive chunks agree
    # within 1.5 % (every rank takes the same decision: MAX over ranks of "not settled yet"), at most FD_BENCH_PRIME_MAX_S (4 s).
    # (the slow state is intermittent: 3 of 5 runs behind an idle GPU with 0.2 s of priming, 0 of 3 with >= 1.5 s)
    prime_log = []
    if not a.mixed_n and float(os.environ.get("FD_BENCH_PRIME_MAX_S", "4")) > 0:
        t_begin = time.perf_counter()
        last = None
        while time.perf_counter() - t_begin < float(os.environ.get("FD_BENCH_PRIME_MAX_S", "4")):
            t1 = time.perf_counter()
            for i in range(4):
                step(i)
            barrier()
            cur = (time.perf_counter() - t1) / 4 * 1e3
            prime_log.append(round(cur, 2))
            n_prime += 4
            settled = last is not None and abs(cur - last) <= 0.015 * min(cur, last)
            last = cur
            flag = torch.tensor([0.0 if settled else 1.0], device=dev)
            if world > 1:
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            if (float(flag.item()) == 0.0 and time.perf_counter() - t_begin >= float(os.environ.get("FD_BENCH_PRIME_MIN_S", "1.5"))
                    and not os.environ.get("FD_BENCH_PRIME_FORCE")):
                break
    for i in range(a.warmup):
        step(i)
    if a.mixed_n:
        # every length of the timed schedule once, untimed: a new length means new kernel variants (first use of a code
        # object), allocator growth and per-length caches -- on a fresh box that was half of a 12-step timed region
        for i in range(a.warmup, a.warmup + a.steps):
            step(i)
    barrier()
    # the interpreter's cyclic garbage collector stays out of the timed region (as timeit does): the host runs ~13 ms per step
    # ahead of the GPU, and a full collection over the step's tensor graph in the middle of the K steps can eat that lead.  One HIP
    # event per step (recorded behind the optimiser launch, read after the region) gives the per-step spread as a diagnostic.
    import gc
    gc.collect()
    gc.disable()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    dev_allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)     # hipMalloc calls of the caching allocator so far
    marks[0].record()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    dev_allocs = torch.cuda.memory_stats().get("num_device_alloc", 0) - dev_allocs0
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    slowest_step = max(range(a.steps), key=lambda i: per_step[i])
    if os.environ.get("FD_BENCH_STEP_TRACE"):
        sys.stderr.write("per-step ms (HIP events, in order): " + " ".join(f"{x:.2f}" for x in per_step) + "\n")
    per_step = sorted(per_step)
    if os.environ.get("FD_BENCH_ENQUEUE"):
        # host-side cost of a step: time inside step() with the GPU running asynchronously behind it.  If it is close to
        # ms_per_step the launch stream, not the GPU, bounds the step.
        cpu = []
        for i in range(5):
            t1 = time.perf_counter()
            step(a.warmup + i)
            cpu.append((time.perf_counter() - t1) * 1e3)
        barrier()
        sys.stderr.write(f"host enqueue time per step (ms): {[round(c, 2) for c in cpu]}; timed {dt / a.steps * 1e3:.2f}\n")
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank_ms = [round(dt / a.steps * 1e3, 3)]
    if world > 1:
        allt = [torch.zeros_like(tmax) for _ in range(world)]
        torch.distributed.all_gather(allt, tmax)
        per_rank_ms = [round(float(t.item()) / a.steps * 1e3, 3) for t in allt]
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    residues = sum(n * b for n, b in sched[a.warmup:])
    # the reference's training step with its 50 % self-conditioning forward (every other step here), same shapes
    sc_ms = None
    if a.mode == "train" and not a.mixed_n and not os.environ.get("FD_BENCH_PROFILE"):   # (profiler runs: the plain step only)
        k = max(2, a.steps - a.steps % 2)
        barrier()
        t0 = time.perf_counter()
        for i in range(k):
            step(i, self_cond=(i % 2 == 0))
        barrier()
        sc_ms = (time.perf_counter() - t0) / k * 1e3
    # Per-launch HIP events are NOT recorded inside the timed region: 1200 event records per step cost ~2.5 ms of it,
    # and with the weight-gradient GEMMs on a second stream a launch's start/stop events would span the kernels it shares
    # the GPU with.  The roofline of the dominant kernel is taken from three more steps right after it, same shapes and
    # data, with that side stream switched off (launches serialised) and events around every profiled launch on the
    # stream it runs on.
    from se3_diffusion_amd import ops as fops
    side_was = fops.set_grad_stream(False)
    lib.gemm_profile = []
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof, lib.gemm_profile = lib.gemm_profile, None
    fops.set_grad_stream(side_was)
    # the same step with every GEMM on the exact fp32 MFMA (bitwise fmaf-chain products: FD_GEMM_EXACT_F32=1; the fused
    # edge-transition kernel off), so the line carries both arithmetic choices; not part of `value`
    exact_ms = None
    if not os.environ.get("FD_BENCH_PROFILE") and not a.mixed_n:   # (rocprofv3 runs: keep the trace to the shipped arithmetic)
        was_exact = lib.set_exact_f32(True)
        step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        barrier()
        exact_ms = (time.perf_counter() - t0) / 3 * 1e3
        lib.set_exact_f32(was_exact)

    if rank != 0:
        return
    if os.environ.get("FD_BENCH_GEMM_DETAIL"):
        shapes = {}
        for p in prof:
            s = shapes.setdefault((p[0], p[1], p[2]) + p[6], [0.0, 0.0, 0])
            s[0] += p[3]; s[1] += p[4].elapsed_time(p[5]) * 1e-3; s[2] += 1
        rows = sorted(shapes.items(), key=lambda kv: -kv[1][1])
        sys.stderr.write("tile akc bkc (M,N,K,batch,gate,beta,pair,ksplit)  calls/step  ms/step  TF/s\n")
        for k, v in rows[:80]:
            sys.stderr.write(f"{k}  {v[2] / 3:.1f}  {v[1] / 3 * 1e3:.3f}  {v[0] / v[1] / 1e12:.1f}\n")
    tile, dflops, dtime, dn, all_flops, tot_t, dshape, by_tile = dominant_kernel(prof)
    kname, peak = _KERNELS[tile]
    achieved = dflops / dtime / 1e12
    nprof = 3
    ms = dt / a.steps * 1e3
    traffic = pmc_traffic(tile, dshape[0])
    workload = (f"config/base.yaml ScoreNetwork ({a.blocks} IPA blocks, 17.4M params), per-GPU batch "
                + (f"B={B} x N={N} residues" if not a.mixed_n else
                   "of same-length backbones, N ~ U{100..512} per step shared by all ranks, B = min(32, 5e5 // N^2) "
                   "(BASELINE configs[3])")
                + (", fwd + fused DSM loss + bwd + RCCL grad all-reduce + Adam" if a.mode == "train" else ", forward only"))
    res = {
        "metric": "residues/sec IPA fwd+bwd" if a.mode == "train" else "residues/sec IPA fwd",
        "value": round(world * residues / dt, 1), "unit": "residues/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "priming_steps": n_prime, "priming_chunk_ms": prime_log,
        "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "batches": "one pre-generated batch per step of the schedule" if a.mixed_n else
                              "one HBM-resident synthetic batch per rank, reused by every step (priming, warm-up and timed)", "parallelism": f"dp{world}", "global_batch": world * B, "n_res": N,
                   "ranks": world, "collective_backend": (torch.distributed.get_backend() if world > 1 else None),
                   "dp_overlap": overlap is not None,
                   "rccl_ranks": world if (world > 1 and torch.distributed.get_backend() == "nccl") else 0,
                   "ms_per_step_by_rank": per_rank_ms,
                   "step_ms_spread": {"min": round(per_step[0], 3), "median": round(per_step[len(per_step) // 2], 3),
                                      "max": round(per_step[-1], 3), "slowest_step": slowest_step,
                                      "device_allocations_in_timed_region": int(dev_allocs),
                                      "note": "HIP events between consecutive optimiser launches of the timed steps (rank 0); step 0 "
                                              "starts on an empty queue behind the barrier (the host enqueues ~8 ms per step), so it is "
                                              "the slow one of every run (~ +2.5 ms); device_allocations = hipMalloc calls of torch's "
                                              "caching allocator inside the timed region (0 = every buffer came out of the cache)"},
                   "scaling_curve": "not measured by this line (one N per invocation; the driver composes 1/2/4/8)",
                   "ms_per_step_exact_f32_gemms": None if exact_ms is None else round(exact_ms, 3),
                   "self_conditioning_50pct": None if sc_ms is None else {
                       "ms_per_step": round(sc_ms, 3), "residues_per_s": round(world * B * N / sc_ms * 1e3, 1),
                       "note": "every other step runs the reference's extra no-grad forward (train_se3_diffusion.py:535-537)"},
                   "arithmetic": ARITH},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4),
                     "traffic": None if traffic is None else traffic["bytes_per_launch"], "kernel": kname,
                     "vs_fp32_mfma_peak": round(achieved / 157.3, 4),
                     "frac_of_sustained_peak": None if peak < 400 else round(achieved / SUSTAINED_SPLIT_PEAK, 4),
                     "sustained_peak": None if peak < 400 else SUSTAINED_SPLIT_PEAK,
                     "sustained_peak_note": "v_mfma_f32_32x32x16_bf16 on register operands, every SIMD busy, random data: 1.76 of the "
                                            "nominal 2.5 PFLOP/s (power-limited; tools/probes/mfma_peak.hip), / 6 products",
                     "traffic_detail": traffic,
                     "traffic_note": None if traffic is None else
                     "HBM bytes per launch, launch-weighted over the variants of this kernel the step runs (rocprofv3 --pmc FETCH_SIZE "
                     "/ WRITE_SIZE in separate passes, calibrated on known 1 GiB transfers: tools/pmc_roofline.sh)",
                     "measured_on": "3 steps right after the timed region with the gradient side stream off "
                                    "(HIP events per profiled launch on its stream)",
                     "launches_per_step": dn // nprof,
                     "avg_launch_us": round(dtime / max(1, dn) * 1e6, 2),
                     "algorithmic_flops_per_launch": round(dflops / max(1, dn), 1),
                     "serialised_ms_per_step_by_kernel_class": {str(k): round(v / nprof, 3) for k, v in by_tile.items()},
                     "serialised_gemm_ms_per_step": round(tot_t / nprof * 1e3, 3),
                     "all_gemm_tflops": round(all_flops / max(tot_t, 1e-9) / 1e12, 2),
                     "step_model_tflops": round(all_flops / nprof / (dt / a.steps) / 1e12, 2)},
    }
    if world == 1 and a.mode == "train" and not a.no_sampling and not a.mixed_n and not os.environ.get("FD_BENCH_PROFILE"):
        # BASELINE configs[3] on this GPU inside the same driver-timed command: 12 steps of the mixed-length schedule
        # (dist.mixed_length_schedule: N ~ U{100..512}, B = min(32, 5e5 // N^2)); every length runs once untimed first (new
        # lengths = first use of kernel variants + allocator growth)
        try:
            m_warm, m_steps = 3, 12
            m_sched = fdist.mixed_length_schedule(m_warm + m_steps)
            data[:] = [make_batch(n, b, 1000 * i + rank) for i, (n, b) in enumerate(m_sched)]
            for i in range(m_warm + m_steps):
                step(i)
            barrier()
            t0 = time.perf_counter()
            for i in range(m_steps):
                step(m_warm + i)
            barrier()
            m_dt = time.perf_counter() - t0
            m_res = sum(n * b for n, b in m_sched[m_warm:])
            res["config"]["mixed_n"] = {
                "residues_per_s": round(m_res / m_dt, 1), "steps": m_steps, "ms_per_step": round(m_dt / m_steps * 1e3, 3),
                "schedule_n_b": [list(x) for x in m_sched[m_warm:]],
                "note": "BASELINE configs[3] on one GPU: the full training step on a seeded mixed-length schedule, N ~ U{100..512}, "
                        "B = min(32, 5e5 // N^2); `python bench.py --mixed-n` is the same measurement as its own line"}
        except Exception as e:  # noqa: BLE001 -- must not lose the training measurement
            res["config"]["mixed_n"] = {"error": repr(e)}
    if world == 1 and a.mode == "train" and not a.no_sampling and not a.mixed_n:
        try:
            data.clear()
            torch.cuda.empty_cache()
            res["config"]["sampling"] = dict(
                sampling_rates(dev, a.blocks, a.sample_steps, lib),
                note=f"backbones/s of the 500-step reverse diffusion (501 network forwards), device-resident loop, one hipGraph per "
                     f"step; N=128 / N=256 at B=1 are FULL 500-step trajectories, the other cases bounded runs of {a.sample_steps} "
                     f"steps scaled to 501 forwards")
        except Exception as e:  # noqa: BLE001 -- must not lose the training measurement
            res["config"]["sampling"] = {"error": repr(e)}
    if not a.no_cpu_baseline and world == 1 and not a.mixed_n:
        try:
            res["cpu_baseline"] = cpu_baseline(N, a.blocks, a.cpu_sample_batch)
        except Exception as e:  # noqa: BLE001 -- the baseline must not lose the GPU measurement
            res["cpu_baseline"] = {"value": None, "error": repr(e)}
        if a.mode == "train" and not a.no_sampling:
            try:
                res["cpu_baseline"]["sampling"] = cpu_sampling_baseline(128, a.blocks)
            except Exception as e:  # noqa: BLE001
                res["cpu_baseline"]["sampling"] = {"value": None, "error": repr(e)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
