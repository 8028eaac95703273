"""Parity at the sizes the product ships at (BASELINE.json configs 2 / 3 / 5), gfx950 only (-m gpu).

Two checkers, both at FULL depth (4 blocks):
  * the oracle (oracle/framediff_oracle.py) run on the GPU box's host cores on the same seeded inputs -- outputs and all
    282 parameter gradients at B=2 x N=128, B=1 x N=256 and at the benchmarked step itself (B=30 x N=128 through the
    module + FlatAdam layout + fused loss, side stream on), outputs at B=1 x N=512;
  * fixtures written by the UNMODIFIED reference in the build container (oracle/make_golden_full.py):
    tests/golden/fwd_n128_b2.npz, fwd_n256_b1.npz (outputs + gradient signatures), fwd_n512_b1.npz (outputs),
    traj_n128.npz, traj_n256.npz, traj_n512_b2.npz (5 reverse steps of Experiment.inference_fn at B=1 x N=128 / 256 and
    B=2 x N=512 with the reference's numpy draws injected; eager and one hipGraph per step).

Every case runs three times: with the shipped GEMM selection, with the persistent split-bf16 kernel forced on
(fd_gemm_set_persistent_blocks(8): the B=30 configuration's kernel at a size the oracle can check; at N=256 the
M >= 65536 weight-gradient tiles 4/6 engage by themselves) and with FD_GEMM_EXACT_F32 (every GEMM a bitwise fp32
fmaf chain; the fused split-bf16 edge-transition kernel is then replaced by the unfused fp32 launch sequence).

Tolerances (fp32, the table in DESIGN.md "Numerics"; written in tests/test_network.py next to the measured values they are <= 10 x
of): outputs per key 1e-5 ... 8e-4 of the tensor's max magnitude, parameter gradients 2e-3 of the tensor's max magnitude (2e-4 /
1.2e-3 for two families) + 2e-5 absolute (analytically-zero gradients) and 2e-3 relative L2 per tensor; a ReLU-fed Linear may
show at most two flipped hidden units (rows of its weight / entries of its bias) within 1e-2, counted and bounded per case
(test_network.grad_mismatch / check_kinks).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import framediff_oracle as fo  # noqa: E402
from se3_diffusion_amd import trunk  # noqa: E402
from test_network import relerr, quat_align, grad_mismatch, check_kinks, check_out  # noqa: E402
from test_network import TOL_OUT, TOL_ROT, TOL_GRAD, ABS_GRAD, TOL_GRAD_L2  # noqa: E402
import parity_log  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
OUT_KEYS = ["rot_score", "trans_score", "rigids", "atom37", "psi"]
MODES = ("shipped", "persistent", "exact_f32")
TOL_LOSS = 1e-4
# 50 chained steps: measured 1.4e-5 (rotation matrices), 1.2e-4 A (translations), 1e-6 (psi), eager and graph-replayed alike
TOL_TRAJ50 = (1.5e-4, 1.5e-3, 1e-4)
# the metric's own schedule (500 steps, dt = 1/500): 499 chained reverse steps stay within 1.6e-5 (rotation matrices) / 1.4e-4 A
# (translations); the LAST step is the network's own frame prediction (train_se3_diffusion.py:778-780), a forward that amplifies a
# perturbation of its input frames 6-8 x: measured 1.4e-3 / 1.6e-3 A there (6 of 128 residues above 1e-4, median 1e-5), psi 6.4e-5;
# eager = graph-replayed (profiles/r05_traj500_error_growth.txt, tools/probes/traj500_probe.py)
TOL_TRAJ500 = (5e-3, 1.5e-2, 6e-4)
# ... and the 499 chained reverse steps on their own (every compared step but the last): 3 x the measured 1.6e-5 / 1.4e-4 A
TOL_TRAJ500_CHAIN = (5e-5, 5e-4)
TOL_GSIG = 2e-3       # gradient signatures (sum, norm) of the reference's large tensors


class gemm_mode:
    """shipped | persistent (8 persistent blocks: engages gemm_bx3p_kernel from 16 tiles up) | exact_f32"""

    def __init__(self, lib, mode):
        self.lib, self.mode = lib, mode

    def __enter__(self):
        c = self.lib.cdll
        self.was_p = c.fd_gemm_set_persistent_blocks(8 if self.mode == "persistent" else 256)
        self.was_e = self.lib.set_exact_f32(self.mode == "exact_f32")

    def __exit__(self, *a):
        self.lib.cdll.fd_gemm_set_persistent_blocks(self.was_p)     # the setters return the previous value
        self.lib.set_exact_f32(self.was_e)


def _check_outputs(out, ref):
    errs = {}
    for k in ["psi", "trans_score", "atom37", "atom14"]:
        check_out(k, out[k], ref[k], TOL_OUT, errs=errs)
    check_out("rot_score", out["rot_score"], ref["rot_score"], TOL_ROT, errs=errs)
    rr = torch.as_tensor(ref["rigids"]).detach()
    check_out("rigids", quat_align(out["rigids"].cpu(), rr), rr, TOL_OUT, errs=errs)
    return errs


_ORACLE = {}


def _oracle(B, N, seed, n_pad, n_fixed, grad):
    """oracle outputs (+ gradients of the fixed random projection) on the host, once per case"""
    key = (B, N, seed, n_pad, n_fixed, grad)
    if key not in _ORACLE:
        conf = dict(fo.CONF, num_blocks=4)
        P = fo.synth_params(seed=seed, conf=conf)
        feats = fo.synth_feats(B, N, seed=seed, n_pad=n_pad, n_fixed=n_fixed)
        torch.set_num_threads(min(32, os.cpu_count() or 8))
        if grad:
            Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            ref = fo.score_network_forward(Po, feats, conf, tfmr_mask_mode="additive")
            rs = np.random.RandomState(77 + seed)
            wts = {k: torch.tensor(rs.standard_normal(tuple(ref[k].shape))).to(ref[k].dtype) for k in OUT_KEYS}
            sum((ref[k] * wts[k]).sum() for k in wts).backward()
            grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Po.items()}
            ref = {k: v.detach() for k, v in ref.items()}
        else:
            with torch.no_grad():
                ref = fo.score_network_forward(P, feats, conf, tfmr_mask_mode="additive")
            wts, grads = None, None
        _ORACLE[key] = (P, feats, ref, wts, grads)
    return _ORACLE[key]


def _run_vs_oracle(lib, mode, B, N, seed, n_pad=0, n_fixed=0, grad=True):
    with parity_log.case(f"oracle B={B} N={N} {mode} grad={grad}"):
        _run_vs_oracle_body(lib, mode, B, N, seed, n_pad, n_fixed, grad)


def _run_vs_oracle_body(lib, mode, B, N, seed, n_pad, n_fixed, grad):
    P, feats, ref, wts, grads = _oracle(B, N, seed, n_pad, n_fixed, grad)
    Pd = {k: v.cuda() for k, v in P.items()}
    fd = {k: v.cuda() for k, v in feats.items()}
    with gemm_mode(lib, mode):
        out, sv = trunk.forward(Pd, fd, 4, save=grad)
        _check_outputs(out, ref)
        if not grad:
            return
        G = {k: torch.zeros_like(v) for k, v in Pd.items()}
        trunk.backward(Pd, G, sv, {k: v.cuda() for k, v in wts.items()})
    bad, kinks = [], []
    for k, g_ref in grads.items():
        mm = grad_mismatch(G[k], g_ref, tol=TOL_GRAD, floor=ABS_GRAD, name=k, kinks=kinks, l2_tol=TOL_GRAD_L2)
        if mm is not None:
            bad.append((k,) + mm)
    assert not bad, (mode, bad[:10])
    check_kinks(kinks, f"oracle B={B} N={N} {mode}")


@pytest.mark.parametrize("mode", MODES)
def test_oracle_n128_b2_full_depth(hip_lib, mode):
    """config 2's kernel selection at an oracle-checkable batch: forward + all 282 gradients"""
    _run_vs_oracle(hip_lib, mode, B=2, N=128, seed=51, n_pad=4, n_fixed=3)


@pytest.mark.parametrize("mode", MODES)
def test_oracle_n256_b1_full_depth(hip_lib, mode):
    """config 3's size: forward + all gradients (M = 65536 pair rows: split-bf16 dW tiles 4 / 6 engage)"""
    _run_vs_oracle(hip_lib, mode, B=1, N=256, seed=52)


@pytest.mark.parametrize("mode", ("shipped", "exact_f32"))
def test_oracle_n512_b1_forward(hip_lib, mode):
    """config 5's length: forward"""
    _run_vs_oracle(hip_lib, mode, B=1, N=512, seed=53, n_pad=9, grad=False)


def test_oracle_n512_b1_all_gradients(hip_lib):
    """BASELINE configs[3] trains at N up to 512 (experiments/train_se3_diffusion.py:524-693 on cluster_time_batch lengths):
    forward + all 282 parameter gradients at B=1 x N=512 against the oracle (262,144 pair rows: 4096 tiles of the fused edge
    kernels on 512 blocks = the dynamic hand-out, split-bf16 dW tiles, N=512 attention kernels without the LDS image of zb)."""
    from se3_diffusion_amd import ops
    n0 = ops.STATS["edge_dynamic_launches"]
    _run_vs_oracle(hip_lib, "shipped", B=1, N=512, seed=54, n_pad=5, n_fixed=4)
    assert ops.STATS["edge_dynamic_launches"] > n0


def test_n512_b8_forward_batch_of_verified_examples(hip_lib):
    """BASELINE configs[4]'s batch (B=8 x N=512, the sampling stress case): the forward of the batch against the forward of each
    example alone -- and example 0 alone IS the oracle-verified B=1 case above (same seed), so every example of the batch is
    tied to the oracle through the same kernels at B=1.  Bound: 2e-5 of the tensor maximum (batching only changes tile
    boundaries and reduction order, not arithmetic)."""
    B, N, seed = 8, 512, 53
    conf = dict(fo.CONF, num_blocks=4)
    P = {k: v.cuda() for k, v in fo.synth_params(seed=seed, conf=conf).items()}
    # example 0 = the oracle-checked inputs of test_oracle_n512_b1_forward; the others are fresh draws
    parts = [fo.synth_feats(1, N, seed=seed + 100 * b, n_pad=9 if b == 0 else b) for b in range(B)]
    feats = {k: torch.cat([p[k] for p in parts], 0).cuda() for k in parts[0]}
    with parity_log.case("batch B=8 N=512 vs per-example forwards"):
        with torch.no_grad():
            big, _ = trunk.forward(P, feats, 4, save=False)
            for b in (0, 3, 7):
                one, _ = trunk.forward(P, {k: v[b:b + 1].contiguous() for k, v in feats.items()}, 4, save=False)
                for k in ["psi", "trans_score", "atom37", "rot_score"]:
                    check_out(k, big[k][b:b + 1], one[k], 2e-5)
                check_out("rigids", quat_align(big["rigids"][b:b + 1].cpu(), one["rigids"].cpu()), one["rigids"].cpu(), 2e-5)
    # and example 0 against the oracle itself
    _P, _f, ref, _w, _g = _oracle(1, N, seed, 9, 0, False)
    _check_outputs({k: v[0:1] for k, v in big.items()}, ref)


def _module_step_vs_oracle(hip_lib, B, N, seed, label, min_dynamic=0):
    """One training step through the ScoreNetwork module with the FlatAdam(adjacent=flat_layout_groups()) layout (merged IPA
    projections, [linear_b ; down_z] as one matrix), accumulate_into_grad, gradient side stream ON, fused device loss -- outputs,
    loss and all 282 parameter gradients against the oracle (torch-CPU restatement on the box's host cores).  Returns the
    launch-path counters the step moved (ops.STATS)."""
    from se3_diffusion_amd import loss as floss, ops, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    from se3_diffusion_amd.optim import FlatAdam
    blocks = 4
    conf = dict(fo.CONF, num_blocks=blocks)
    P = fo.synth_params(seed=seed, conf=conf)
    model = ScoreNetwork(ts.base_model_conf(blocks), diffuser=None)
    model.load_state_dict(P, strict=True)
    model = model.cuda().train()
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=model.flat_layout_groups())
    model.accumulate_into_grad = True
    batch = ts.synthetic_batch(B, N, "cuda", seed=100 + seed)
    batch["t"][:5] = torch.tensor([0.03, 0.15, 0.22, 0.6, 0.99], device="cuda")     # both sides of every loss threshold
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])
    assert ops.side_active(batch["rigids_t"], B * N * N), "the benchmarked step runs its weight gradients on the side stream"
    was_p = hip_lib.cdll.fd_gemm_set_persistent_blocks(256)
    assert was_p == 256 and not hip_lib.exact_f32
    from se3_diffusion_amd import network as nw
    assert nw._proj_views(dict(model.named_parameters()), "score_model.trunk.ipa_0") is not None   # merged projections
    before = dict(ops.STATS)
    for _ in range(2):            # second pass = the steady state the benchmark times (allocator, cached views)
        opt.zero()
        out = model(batch)
        loss = floss.dsm_loss(batch, out, gt37)
        loss.backward()
    torch.cuda.synchronize()
    moved = {k: ops.STATS[k] - before[k] for k in before}
    assert moved["edge_dynamic_launches"] >= min_dynamic, moved
    cpu_batch = {k: v.cpu() for k, v in batch.items()}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = fo.score_network_forward(Po, cpu_batch, conf, tfmr_mask_mode="additive")
    lref = ts.dsm_loss(cpu_batch, ref, gt37.cpu())
    lref.backward()
    with parity_log.case(label):
        _check_outputs({k: v.detach() for k, v in out.items()}, {k: v.detach() for k, v in ref.items()})
        lerr = abs(float(loss) - float(lref)) / abs(float(lref))
        parity_log.out("loss", lerr)
        assert lerr < TOL_LOSS, (float(loss), float(lref))
        bad, kinks = [], []
        for n, p in model.named_parameters():
            g_ref = Po[n].grad if Po[n].grad is not None else torch.zeros_like(Po[n])
            mm = grad_mismatch(p.grad, g_ref, tol=TOL_GRAD, floor=ABS_GRAD, name=n, kinks=kinks, l2_tol=TOL_GRAD_L2)
            if mm is not None:
                bad.append((n,) + mm)
        assert not bad, bad[:10]
        check_kinks(kinks, label)
    return moved


def test_oracle_n128_b30_benchmarked_step(hip_lib):
    """The EXACT launch configuration bench.py times (BASELINE configs[1]): B=30 x N=128, full depth, 160-block pair_dw beside the
    main stream, >= 512-tile persistent split GEMMs (~20 GB / ~20 s of oracle on the box's host cores).
    491,520 pair rows = 7,680 tiles on 512 blocks: every fused edge launch (3 fwd + 3 bwd) and the embedder's backward hand their
    tiles out through the atomic counter (FdEdgeMlpDesc.sched != null) -- the path the benchmark times; 240 query tiles: the
    one-launch IPA attention kernels in both directions."""
    moved = _module_step_vs_oracle(hip_lib, 30, 128, 61, "oracle B=30 N=128 benchmarked step (module + FlatAdam + fused loss, side stream on)",
                                   min_dynamic=2 * 7)
    assert moved["ipa_flash_fwd"] == 8 and moved["ipa_flash_bwd"] == 8 and moved["ipa_sequence_bwd"] == 0, moved


def test_oracle_n200_b12_mixed_length_step(hip_lib):
    """A configs[3] shape (dist.mixed_length_schedule: N ~ U{100..512}, B = min(32, 5e5 // N^2); reference
    experiments/train_se3_diffusion.py:524-693, data/utils.py:387-399) that takes the one-launch IPA attention in BOTH directions
    with a RAGGED last query / key tile: B=12 x N=200 = 156 query tiles, N % 16 = 8.  480,000 pair rows: 3,750 tiles of the
    8-wave edge kernel with a ragged last tile.  All 282 gradients against the oracle."""
    moved = _module_step_vs_oracle(hip_lib, 12, 200, 62, "oracle B=12 N=200 mixed-length step (flash IPA fwd + bwd, ragged tiles)")
    assert moved["ipa_flash_fwd"] == 8 and moved["ipa_flash_bwd"] == 8 and moved["ipa_sequence_bwd"] == 0, moved


def test_oracle_n256_b7_mixed_length_step(hip_lib):
    """The other mixed-length regime: B=7 x N=256 = 112 query tiles -- the one-launch IPA forward (>= 80 tiles) followed by the
    GEMM-sequence backward (< 128 tiles: fd_ipa_attn_bwd reads the key-point copy the forward decided to keep).  All 282
    gradients against the oracle."""
    moved = _module_step_vs_oracle(hip_lib, 7, 256, 63, "oracle B=7 N=256 mixed-length step (flash IPA fwd, sequence bwd)")
    assert moved["ipa_flash_fwd"] == 8 and moved["ipa_flash_bwd"] == 0 and moved["ipa_sequence_bwd"] == 8, moved


def _golden_full(lib, name, mode):
    with parity_log.case(f"reference golden {name} {mode}"):
        _golden_full_body(lib, name, mode)


def _golden_full_body(lib, name, mode):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, N, seed = int(g["B"]), int(g["N"]), int(g["seed"])
    conf = dict(fo.CONF, num_blocks=int(g["blocks"]))
    P = {k: v.cuda() for k, v in fo.synth_params(seed=seed, conf=conf).items()}
    feats = {k: v.cuda() for k, v in fo.synth_feats(B, N, seed=seed, n_pad=int(g["n_pad"]), n_fixed=int(g["n_fixed"])).items()}
    has_grad = "w_seed" in g.files
    with gemm_mode(lib, mode):
        out, sv = trunk.forward(P, feats, int(g["blocks"]), save=has_grad)
        ref = {k: torch.tensor(g["out_" + k]) for k in ["psi", "rot_score", "trans_score", "rigids"]}
        got = dict(out)
        for k in ("atom37", "atom14"):
            ref[k] = torch.tensor(g["out_" + k])
            got[k] = out[k][:, :, :5]
            if k == "atom37":
                assert float(out[k][:, :, 5:].abs().max()) == 0.0
        _check_outputs(got, ref)
        if not has_grad:
            return
        rs = np.random.RandomState(int(g["w_seed"]))
        wts = {}
        for k in OUT_KEYS:
            shp = tuple(out[k].shape)
            wts[k] = torch.tensor(rs.standard_normal(shp)).to(out[k].dtype).cuda()
        G = {k: torch.zeros_like(v) for k, v in P.items()}
        trunk.backward(P, G, sv, wts)
    kinks = []
    for key in g.files:
        if key.startswith("grad/"):
            n = key[5:]
            mm = grad_mismatch(G[n], torch.tensor(g[key]), tol=TOL_GRAD, floor=ABS_GRAD, name=n, kinks=kinks, l2_tol=TOL_GRAD_L2)
            assert mm is None, (n, mm)
        elif key.startswith("gsig/"):
            n = key[5:]
            s, a, l2 = g[key]
            gg = G[n].cpu().double()
            e_norm = abs(float(gg.norm()) - l2) / (l2 + 1e-30)
            e_sum = abs(float(gg.sum()) - s) / (a + 1e-30)
            if l2 > 1e-6:
                parity_log.out("gsig_norm", e_norm)
                parity_log.out("gsig_sum", e_sum)
            assert abs(float(gg.norm()) - l2) < TOL_GSIG * l2 + 1e-6, (n, float(gg.norm()), l2)
            assert abs(float(gg.sum()) - s) < TOL_GSIG * a + 1e-6, n
    check_kinks(kinks, f"{name} {mode}")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["fwd_n128_b2", "fwd_n256_b1"])
def test_reference_golden_full_depth(hip_lib, name, mode):
    """outputs + gradient signatures of the unmodified reference at N=128 (B=2) and N=256"""
    _golden_full(hip_lib, name, mode)


def test_reference_golden_n512(hip_lib):
    """outputs + gradient signatures of the unmodified reference at B=1 x N=512 (backward included since round 4)"""
    _golden_full(hip_lib, "fwd_n512_b1", "shipped")


# 5-step trajectories: measured worst 4.2e-5 (rotation matrices), 9e-4 A (translations, B=2 x N=512), 1e-5 (psi)
def _trajectory(fixture, use_graph, tol_rot=4e-4, tol_trans=9e-3, tol_psi=1e-4, tol_chain=None):
    """tol_chain = (rot, trans): a tighter bound for every compared step except the last one (the network's own frame prediction)"""
    from se3_diffusion_amd import sampler, train_step as ts
    from se3_diffusion_amd.data import se3_diffuser, utils as du
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    from test_diffuser import conf as dconf
    T = np.load(os.path.join(GOLD, fixture + ".npz"))
    diff = se3_diffuser.SE3Diffuser(dconf())
    blocks = int(T["blocks"])
    m = ScoreNetwork(ts.base_model_conf(blocks), diff)
    m.load_state_dict(fo.synth_params(seed=int(T["seed"]), conf=dict(fo.CONF, num_blocks=blocks)), strict=True)
    m = m.cuda().eval()
    B, N = int(T["B"]), int(T["N"])
    feats = sampler.init_feats(diff, B, N, "cuda", noise=(T["init_randn"], T["init_rand"], T["init_normal"]))
    if "rng_keys" in T:
        # the 500-step fixture stores the state of numpy's global generator at the reference's first diffuser.reverse call instead
        # of 3 MB of draws (rotation first, then translation, se3_diffuser.py:213-262; sampler.sample asks for them in step order)
        rs = np.random.RandomState()
        rs.set_state(("MT19937", T["rng_keys"], int(T["rng_pos"]), int(T["rng_has_gauss"]), float(T["rng_cached"])))
        noise_fn = lambda i, shp: (rs.normal(size=shp), rs.normal(size=shp))
        step_index = [int(i) for i in T["step_index"]]
    else:
        zr, zt = T["z_rot"], T["z_trans"]
        noise_fn = lambda i, shp: (zr[i], zt[i])
        step_index = list(range(len(T["step_rigids"])))
    out = sampler.sample(m, diff, feats, num_t=int(T["num_t"]), min_t=float(T["min_t"]), noise_scale=float(T["noise_scale"]),
                         noise_fn=noise_fn, return_traj=True, use_graph=use_graph)
    rm = lambda q: du.quat_wxyz_to_matrix(np.asarray(q)[..., :4].astype(np.float64))
    r0 = feats["rigids_t"].cpu().numpy()                    # same starting frames (quaternions up to sign)
    assert np.abs(rm(r0) - rm(T["rig_init"])).max() < 1e-5 and np.abs(r0[..., 4:] - T["rig_init"][..., 4:]).max() < 1e-4
    growth = []
    with parity_log.case(f"trajectory {fixture} graph={use_graph}"):
        assert len(out["rigid_traj"]) == int(T["num_t"])
        for i, ref in zip(step_index, T["step_rigids"]):
            got = out["rigid_traj"][i].cpu().numpy()
            er, et = np.abs(rm(got) - rm(ref)).max(), np.abs(got[..., 4:] - ref[..., 4:]).max()
            growth.append((float(f"{er:.2e}"), float(f"{et:.2e}")))
            parity_log.out("rot_matrix_abs", er)
            parity_log.out("trans_angstrom_abs", et)
            assert er < tol_rot, (i, growth)
            assert et < tol_trans, (i, growth)          # Angstrom (coordinates of +-30 A)
            if tol_chain is not None and i != step_index[-1]:
                parity_log.out("chain_rot_matrix_abs", er)
                parity_log.out("chain_trans_angstrom_abs", et)
                assert er < tol_chain[0] and et < tol_chain[1], (i, growth)
        ep = np.abs(out["psi"].cpu().numpy() - T["final_psi"]).max()
        parity_log.out("psi_abs", ep)
        assert ep < tol_psi
        if len(growth) > 8:
            pick = sorted({0, 4, 9, 24, len(growth) // 2, (3 * len(growth)) // 4, (9 * len(growth)) // 10, len(growth) - 2, len(growth) - 1})
            print(f"[parity] {fixture} graph={use_graph}: (rot, trans A) error at compared steps "
                  f"{[step_index[j] + 1 for j in pick if j < len(growth)]}: {[growth[j] for j in pick if j < len(growth)]}")


@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_trajectory_n128(hip_lib, use_graph):
    """5 reverse steps of the reference's inference loop at N=128, full depth, reference noise injected
    (experiments/train_se3_diffusion.py:746-781)"""
    _trajectory("traj_n128", use_graph)


@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_trajectory_n256(hip_lib, use_graph):
    """BASELINE configs[2]'s length: 5 reverse steps of the UNMODIFIED Experiment.inference_fn at B=1 x N=256 (fixture written
    by oracle/make_golden_full.py::traj_via_experiment), eager and one-hipGraph-per-step"""
    _trajectory("traj_n256", use_graph)


@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_trajectory_n512_b2(hip_lib, use_graph):
    """BASELINE configs[4]'s length, batched: 5 reverse steps of the unmodified Experiment.inference_fn at B=2 x N=512 (per-
    example centring, the eigh-free frame path and the batched kernels of the N=512 sampling configuration)"""
    _trajectory("traj_n512_b2", use_graph)


@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_trajectory_n128_50_steps(hip_lib, use_graph):
    """Error growth over a tenth of the metric's 500-step trajectory: 50 reverse steps (51 forwards) of the UNMODIFIED
    Experiment.inference_fn at B=1 x N=128 (fixture traj_n128_t50, oracle/make_golden_full.py::traj_via_experiment), the same
    noise injected, eager and hipGraph-replayed.  Every step is compared; the per-step errors are printed."""
    _trajectory("traj_n128_t50", use_graph, tol_rot=TOL_TRAJ50[0], tol_trans=TOL_TRAJ50[1], tol_psi=TOL_TRAJ50[2])


@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_trajectory_n128_500_steps(hip_lib, use_graph):
    """THE METRIC'S SCHEDULE: 500 reverse steps (501 forwards, dt = 1/500, the 500-point t grid -- other t_to_idx rows and another
    g(t) / b(t) sequence than the 5- and 50-step fixtures) of the UNMODIFIED Experiment.inference_fn at B=1 x N=128
    (experiments/train_se3_diffusion.py:746-781; fixture traj_n128_t500, oracle/make_golden_full.py::traj_via_experiment), the
    reference's noise stream injected, eager and hipGraph-replayed.  Every 10th step is compared; error growth is printed."""
    _trajectory("traj_n128_t500", use_graph, tol_rot=TOL_TRAJ500[0], tol_trans=TOL_TRAJ500[1], tol_psi=TOL_TRAJ500[2],
                tol_chain=TOL_TRAJ500_CHAIN)


@pytest.mark.parametrize("use_graph", [False, True])
def test_reference_trajectory_n256_500_steps(hip_lib, use_graph):
    """BASELINE.json configs[2] EXACTLY: reverse-diffusion inference, 500 steps, N=256 -- 500 reverse steps (501 forwards) of the
    UNMODIFIED Experiment.inference_fn at B=1 x N=256 (fixture traj_n256_t500, oracle/make_golden_full.py::traj_via_experiment;
    631 s of reference CPU time), the reference's noise stream injected, eager and hipGraph-replayed; every 10th step compared, the
    chained reverse steps against their own tight bound, the last step (the network's frame prediction) separately."""
    _trajectory("traj_n256_t500", use_graph, tol_rot=TOL_TRAJ500[0], tol_trans=TOL_TRAJ500[1], tol_psi=TOL_TRAJ500[2],
                tol_chain=TOL_TRAJ500_CHAIN)
