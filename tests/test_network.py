"""ScoreNetwork forward/backward parity: HIP kernels vs the oracle (oracle/framediff_oracle.py).

CPU tier: kernel sources under the SIMT interpreter (tests/emu) at small N.
GPU tier (-m gpu): the gfx950 library, same checks plus larger N and the golden fixtures.

Tolerances (fp32): outputs rtol 2e-4 of the tensor's max magnitude (rot_score 1e-3: the
reference itself evaluates the IGSO(3) series with float32 sin/cos arguments, see
DESIGN.md "numerics"); parameter gradients 2e-3 of max magnitude with an absolute floor for
gradients that are analytically zero (linear_b.bias: softmax shift invariance).
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
import parity_log  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def relerr(a, b, floor=1e-12):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + floor))


# Per-output bounds = <= 10 x the worst value any GPU parity case achieved in round 4 (profiles/r04_parity_errors.md: 35 cases,
# N = 12 ... 512, B up to 30, three GEMM modes): trans_score 2.9e-6, atom37 / atom14 7.3e-6, rigids 6.1e-7, rot_score 7.6e-5 (the
# B=30 step; 1e-6 typical), psi 1.4e-4 in ONE case whose torsion pre-activation is nearly the zero vector (B=3 x N=37; 1e-6 ... 1e-5
# everywhere else) -- psi keeps the 2e-4 it had.  The same errors appear with every GEMM a bitwise fp32 chain (exact_f32 mode): they
# are the fp32 oracle's own summation order, not the split-bf16 arithmetic.
TOL_OUT_KEY = {"trans_score": 3e-5, "atom37": 7e-5, "atom14": 7e-5, "rigids": 1e-5, "rot_score": 8e-4, "psi": 2e-4}


def check_out(key, got, ref, tol, floor=1e-12, errs=None):
    """max |got - ref| / max |ref| < tol (the tighter of the caller's bound and TOL_OUT_KEY's), with the achieved value recorded
    (parity_log) and returned"""
    tol = min(tol, TOL_OUT_KEY.get(key, tol))
    e = relerr(got, ref, floor)
    parity_log.out(key, e)
    if errs is not None:
        errs[key] = e
    assert e < tol, (key, e, tol, errs)
    return e


# Parameters whose gradient has entries fed by ONE (row, hidden unit) of a ReLU: the weight / bias of the Linear in front of
# it (dW[unit, :] = sum_rows [h > 0] dh x, db[unit]).  Only these can show an isolated ReLU-kink outlier; everywhere else a
# flipped unit is one of >= 10^4 summands and stays inside the ordinary bound.
_RELU_FED = (r"embedding_layer\.(node|edge)_embedder\.(0|2)\.", r"node_transition_\d+\.linear_(1|2)\.",
             r"edge_transition_\d+\.trunk\.(0|2)\.", r"seq_tfmr_\d+\.layers\.\d+\.linear1\.", r"torsion_pred\.linear_1\.")
MAX_KINKS_PER_CASE = 6


def relu_fed(name):
    import re
    return name is not None and any(re.search(p, name) for p in _RELU_FED)


def grad_mismatch(g, g_ref, tol=2e-3, floor=2e-5, kink_tol=1e-2, kink_units=2, name=None, kinks=None, l2_tol=None):
    """None if `g` matches `g_ref`, else (max error, scale).  Bound: tol * max|g_ref| + floor per element.  ReLU kinks: a
    hidden unit whose pre-activation lies within fp32 round-off of zero switches on / off between two correct fp32
    implementations (measured: the fused and the unfused edge embedder agree to 6e-7 relative, yet one unit of
    edge_transition_0 flips at N=24).  A flipped (row, unit) moves dW[unit, :] of the Linear in front of the ReLU by
    dh[row, unit] * x[row, :] -- ONE ROW of that weight (and one bias entry), nothing else.  Accepted and RECORDED in `kinks`
    (a list the caller owns): offending entries confined to at most `kink_units` rows of a weight / entries of a bias of a
    ReLU-fed Linear (`relu_fed(name)`; with name=None nothing is excused), each within kink_tol * max|g_ref|."""
    g2 = g.detach().double().cpu()
    r2 = g_ref.detach().double().cpu()
    scale = float(r2.abs().max())
    err = (g2 - r2).abs()
    if name is not None:
        tol = min(tol, TOL_GRAD_FAMILY.get(parity_log.family(name), tol))
    over = err > tol * scale + floor
    n_over = int(over.sum())
    if n_over == 0:
        parity_log.grad(name, g2, r2)
        # second figure: relative L2 error of the tensor (a wrong low-magnitude region hides under a max-norm bound)
        if l2_tol is not None and scale > 1e3 * floor:
            l2 = float((g2 - r2).norm()) / (float(r2.norm()) + 1e-30)
            if l2 > l2_tol:
                return l2, float(r2.norm())
        return None
    if relu_fed(name) and float(err.max()) <= kink_tol * scale + floor:
        units = over.reshape(over.shape[0], -1).any(dim=1)          # rows of a weight [out, in] / entries of a bias [out]
        n_units = int(units.sum())
        if n_units <= kink_units:
            if kinks is not None:
                kinks.append((name, n_units, n_over, float(err.max()) / (scale + 1e-30)))
            parity_log.grad(name, g2, r2, excused=True)
            return None
    parity_log.grad(name, g2, r2)
    return float(err.max()), scale


def check_kinks(kinks, what=""):
    """At most MAX_KINKS_PER_CASE excused hidden units (weight rows + bias entries; a flipped unit usually shows in both) over
    all 282 tensors of a case; printed (pytest -s / the failure text)."""
    n = sum(k[1] for k in kinks)
    print(f"[relu kinks] {what}: {n} excused units in {len(kinks)} tensors (name, units, entries, rel err) {kinks}")
    assert n <= MAX_KINKS_PER_CASE, (what, kinks)


def quat_align(a, b):
    s = torch.sign((a[..., :4] * b[..., :4]).sum(-1, keepdim=True))
    return torch.cat([a[..., :4] * s, a[..., 4:]], -1)


# the bounds of the fp32 parity tests (DESIGN.md "Numerics"); tests/parity_log.py records what every case achieves.
# Gradients (max |g - g_ref| / max |g_ref| per tensor): measured worst per parameter family 1.8e-5 (bb_update) ... 4.8e-4 (most
# families, at the B=30 benchmarked step -- a sum over 30 x 128^2 pair rows in fp32 on both sides) ... 1.2e-3 (edge_transition
# trunk.0.weight at N=512, where the oracle itself is 5.5e-4 from the reference): TOL_GRAD = 2e-3 is 1.7 - 4 x those, per-family
# bounds below are tighter where the code achieves more.  TOL_GRAD_L2: relative L2 error of a tensor (worst measured 5.1e-4).
TOL_OUT, TOL_ROT, TOL_GRAD, ABS_GRAD, TOL_GRAD_L2 = 2e-4, 8e-4, 2e-3, 2e-5, 2e-3
# per parameter family: <= 3 x the worst value any GPU case achieves (profiles/r06_parity_errors.md: 9.1e-6 bb_update, 5.4e-5
# torsion_pred, 2.3e-4 .. 3.9e-4 the embedders / IPA / ipa_ln / post_tfmr / skip_embed -- all at the B = 30 benchmarked step, sums
# over 30 x 128 residue rows or 30 x 128^2 pair rows in fp32 on both sides, whose value moves with the tile shapes of the node-level
# GEMMs: 1.4e-4 .. 2.6e-4 with the round-5 tiles); edge_transition / node_transition / seq_tfmr (1.0e-3 .. 1.3e-3 measured, at
# N = 512 and at the B = 30 step) keep TOL_GRAD
TOL_GRAD_FAMILY = {"bb_update": 3e-5, "torsion_pred": 2e-4, "embed.edge": 7e-4, "embed.node": 8e-4, "ipa.head_weights": 8e-4,
                   "ipa.pair_proj": 1.2e-3, "ipa.proj": 1e-3, "ipa_ln": 1e-3, "post_tfmr": 8e-4, "skip_embed": 8e-4}


def run_case(dev, B, N, blocks, seed, n_pad=0, n_fixed=0, check_grad=True, tol_out=TOL_OUT, tol_grad=TOL_GRAD, rot_floor=1e-12,
             tfmr_layers=2):
    conf = dict(fo.CONF, num_blocks=blocks, tfmr_layers=tfmr_layers)
    P = fo.synth_params(seed=seed, conf=conf)
    feats = fo.synth_feats(B, N, seed=seed, n_pad=n_pad, n_fixed=n_fixed)
    Pd = {k: v.to(dev) for k, v in P.items()}
    fd = {k: v.to(dev) for k, v in feats.items()}
    with parity_log.case(f"run_case dev={dev} B={B} N={N} blocks={blocks} seed={seed} pad={n_pad} fixed={n_fixed}"):
        return _run_case(dev, P, feats, Pd, fd, conf, B, N, blocks, seed, check_grad, tol_out, tol_grad, rot_floor)


def _run_case(dev, P, feats, Pd, fd, conf, B, N, blocks, seed, check_grad, tol_out, tol_grad, rot_floor):
    out, sv = trunk.forward(Pd, fd, blocks)
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = fo.score_network_forward(Po, feats, conf, tfmr_mask_mode="additive")
    errs = {}
    for k in ["psi", "trans_score", "atom37", "atom14"]:
        check_out(k, out[k], ref[k], tol_out, errs=errs)
    check_out("rot_score", out["rot_score"], ref["rot_score"], TOL_ROT, rot_floor, errs=errs)
    check_out("rigids", quat_align(out["rigids"].cpu(), ref["rigids"].detach()), ref["rigids"], tol_out, errs=errs)
    if not check_grad:
        return errs
    rs = np.random.RandomState(77 + seed)
    wts = {k: torch.tensor(rs.standard_normal(tuple(ref[k].shape))).to(ref[k].dtype)
           for k in ["rot_score", "trans_score", "rigids", "atom37", "psi"]}
    loss = sum((ref[k] * wts[k]).sum() for k in wts)
    loss.backward()
    G = {k: torch.zeros_like(v) for k, v in Pd.items()}
    trunk.backward(Pd, G, sv, {k: v.to(dev) for k, v in wts.items()})
    bad, kinks = [], []
    for k, v in Po.items():
        g_ref = v.grad if v.grad is not None else torch.zeros_like(v)
        mm = grad_mismatch(G[k], g_ref, tol=tol_grad, floor=ABS_GRAD, name=k, kinks=kinks, l2_tol=TOL_GRAD_L2)
        if mm is not None:
            bad.append((k,) + mm)
    assert not bad, bad[:10]
    check_kinks(kinks, f"B={B} N={N} blocks={blocks} seed={seed}")
    return errs


def test_forward_backward_emu_small(use_emu):
    run_case("cpu", B=1, N=8, blocks=2, seed=3)


def test_degenerate_sizes_emu(use_emu):
    """A single residue, two residues, an example whose residues are all padding, a backbone whose residues are all fixed
    (motif scaffolding's fixed_mask, score_network.py:181-214): forward and every parameter gradient against the oracle."""
    run_case("cpu", B=1, N=1, blocks=1, seed=1)
    run_case("cpu", B=1, N=2, blocks=1, seed=2)
    run_case("cpu", B=2, N=3, blocks=2, seed=3, n_pad=1)
    run_case("cpu", B=2, N=5, blocks=1, seed=4, n_pad=5)
    # all fixed: the frames do not move, the reference's relative rotation is exactly the identity and its rot_score exactly 0;
    # here it is the fp32 round-off of R0^T R0 (|rot_score| ~ 1e-8): compared on an absolute scale
    run_case("cpu", B=1, N=4, blocks=1, seed=5, n_fixed=4, rot_floor=1e-3)


def test_other_transformer_depths_emu(use_emu):
    """model.ipa.seq_tfmr_num_layers other than config/base.yaml's 2 (ipa_pytorch.py:584-593): the layer count is read off the
    state_dict, forward and every gradient against the oracle"""
    run_case("cpu", B=1, N=6, blocks=1, seed=9, tfmr_layers=1)
    run_case("cpu", B=2, N=5, blocks=2, seed=10, n_pad=1, tfmr_layers=3)


@pytest.mark.slow
def test_forward_backward_emu_pad_fixed(use_emu):
    run_case("cpu", B=2, N=9, blocks=2, seed=4, n_pad=2, n_fixed=2)


@pytest.mark.gpu
def test_forward_backward_gpu(hip_lib):
    run_case("cuda", B=2, N=12, blocks=4, seed=0, n_pad=2, n_fixed=3)
    run_case("cuda", B=3, N=40, blocks=4, seed=5)
    run_case("cuda", B=1, N=100, blocks=2, seed=6, n_pad=7, check_grad=False)
    # odd lengths (mixed-length training, BASELINE configs[3]): pair-row counts that are no multiple of 16 / 64 -- ragged
    # last tiles of the fused edge kernels, ragged last stage of the grouped weight gradients, unaligned M of the GEMMs
    run_case("cuda", B=3, N=37, blocks=2, seed=7, n_pad=3)
    run_case("cuda", B=2, N=101, blocks=1, seed=8)


def _golden(dev, name, mode_train=True):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    B, N, seed, blocks = int(g["B"]), int(g["N"]), int(g["seed"]), int(g["blocks"])
    conf = dict(fo.CONF, num_blocks=blocks)
    P = {k: v.to(dev) for k, v in fo.synth_params(seed=seed, conf=conf).items()}
    feats = {k: v.to(dev) for k, v in fo.synth_feats(B, N, seed=seed, n_pad=int(g["n_pad"]), n_fixed=int(g["n_fixed"])).items()}
    with parity_log.case(f"golden {name} dev={dev} train={mode_train}"):
        _golden_body(dev, name, mode_train, g, P, feats, blocks)


def _golden_body(dev, name, mode_train, g, P, feats, blocks):
    out, sv = trunk.forward(P, feats, blocks, tfmr_bool_mask=not mode_train)
    pre = "out_" if mode_train else "eval_"
    for k in ["psi", "trans_score", "atom37", "atom14"]:
        check_out(k, out[k], torch.tensor(g[pre + k]), TOL_OUT)
    check_out("rot_score", out["rot_score"], torch.tensor(g[pre + "rot_score"]), TOL_ROT)
    ref_r = torch.tensor(g[pre + "rigids"])
    check_out("rigids", quat_align(out["rigids"].cpu(), ref_r), ref_r, TOL_OUT)
    if not mode_train:
        return
    wts = {k: torch.tensor(g["w_" + k]).to(dev) for k in ["rot_score", "trans_score", "rigids", "atom37", "psi"]}
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
            assert abs(float(gg.norm()) - l2) < TOL_GRAD * l2 + 1e-6, (n, float(gg.norm()), l2)
            assert abs(float(gg.sum()) - s) < TOL_GRAD * a + 1e-6, n
    check_kinks(kinks, name)


@pytest.mark.gpu
def test_golden_reference_gpu(hip_lib):
    """Against outputs + gradients of the unmodified reference (tests/golden, made by oracle/make_golden.py)."""
    _golden("cuda", "fwd_n12_b2_pad_fixed")
    _golden("cuda", "fwd_n24_b1")
    _golden("cuda", "fwd_n64_b1_1block")
    _golden("cuda", "fwd_n12_b2_pad_fixed", mode_train=False)
