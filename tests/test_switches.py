"""Alternative launch sequences of the same arithmetic give the same training step.  Every field of options.opts that selects
between a fused kernel and the launches it replaces is flipped here (read at call time: no re-import): IPA's four
projections of s as ONE GEMM over back-to-back weights (proj_merge; optim.FlatAdam lays them out so), the sequence-
transformer attention in one launch (fused_seq_attn), the per-row IPA attention kernel (fused_ipa_attn), IPA attention as one launch (flash_ipa), the fused edge
transition + grouped weight gradients (fused_edge, grouped_pair_dw), the grouped node-level weight gradients
(grouped_node_dw), the fused edge embedder (fused_embed), the zero arena,
split-K dX, and the gradient side stream -- model/ipa_pytorch.py:340-374,584-593.

Tolerance: 1e-4 of (each gradient's maximum + 1e-3) -- the merged GEMM has other tile shapes, the fused attention another
summation order, both fp32-accurate (measured 2.1e-5 at B=4 x N=128, the size of either path's distance to the oracle).  linear_b.bias is left out: its gradient is analytically zero (a softmax is invariant
to a shift of its logits) and numerically the round-off of a sum of O(1) terms (|g| ~ 5e-8 at either setting)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import framediff_oracle as fo  # noqa: E402
from se3_diffusion_amd import network as nw, options, train_step as ts  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402
from se3_diffusion_amd.optim import FlatAdam  # noqa: E402


def _step(dev, B, N, blocks, **kw):
    # (0: the fused sequence attention and the one-launch IPA attention at any size)
    with options.override(seq_attn_min_rows=0, flash_ipa_min_tiles=0, flash_ipa_bwd_min_tiles=0, **kw):
        conf = dict(fo.CONF, num_blocks=blocks)
        m = ScoreNetwork(ts.base_model_conf(blocks), diffuser=None)
        m.load_state_dict(fo.synth_params(seed=11, conf=conf), strict=True)
        m = m.to(dev).train()
        opt = FlatAdam(m.parameters(), lr=1e-4, adjacent=m.flat_layout_groups())
        ipa = m.score_model.trunk["ipa_0"]
        assert ipa.linear_q.weight.data_ptr() + 4 * ipa.linear_q.weight.numel() == ipa.linear_kv.weight.data_ptr()
        assert (nw._proj_views(dict(m.named_parameters()), "score_model.trunk.ipa_0") is not None) == options.opts.proj_merge
        batch = ts.synthetic_batch(B, N, dev, seed=5)
        batch["t"][0] = 0.1
        cb = {k: v.cpu() for k, v in batch.items()}
        gt37, _ = fo.backbone_atoms(cb["rigids_0"][..., :4], cb["rigids_0"][..., 4:], cb["torsion_angles_sin_cos"][..., 2, :])
        opt.zero_grad()
        out = m(batch)
        from se3_diffusion_amd import ops
        # (FlatAdam: one flat parameter buffer -> the forward split it, unless the option or exact-fp32 mode says no)
        assert (ops._PLANES is not None) == bool(options.opts.weight_planes and not ops.lib().exact_f32)
        loss = ts.dsm_loss(batch, out, gt37.to(dev))
        loss.backward()
        assert ops._PLANES is None            # (released at the end of the backward)
        return float(loss.detach()), {n: p.grad.detach().double().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}


# (1e-4: the alternative launch sequences sum in different orders; the least well conditioned gradient, the two-entry bias of
# torsion_pred.linear_final -- a sum over all residues with cancellation -- moves by 5e-5 of its size between sequences and by
# 0.3e-5 from run to run of the SAME sequence (atomic accumulation order).)
# one group per comparison: (fields switched OFF together, gradient tolerance).  The fused edge kernels compute in split-bf16
# (fp32-accurate) against fp32 fmaf chains in the unfused sequence: ReLU-kink entries aside, 2e-4.
GROUPS = [
    (dict(proj_merge=False, fused_seq_attn=False), 1e-4),
    (dict(fused_seq_attn_bwd=False), 1e-4),   # fd_seq_attn_bwd vs four batched GEMMs + fd_row_softmax_bwd
    (dict(fused_ipa_attn=False), 1e-4),
    (dict(flash_ipa=False), 1e-4),            # fd_ipa_flash_fwd (probabilities written for the backward) vs the launch sequence
    (dict(flash_ipa_hpb=2), 1e-4),            # (its 2-heads-per-block shape against the default pick)
    (dict(flash_ipa_bwd=False), 1e-4),        # fd_ipa_flash_bwd vs dA GEMMs + fd_ipa_attn_bwd
    (dict(flash_ipa_keys=False), 1e-4),       # fd_ipa_flash_bwd_keys vs the dV / dv_pts / dK GEMMs + fd_ipa_kpts_bwd
    (dict(zero_arena=False, dx_splitk=False, grad_stream=False), 1e-4),
    (dict(grouped_pair_dw=False), 1e-4),
    (dict(grouped_node_dw=False), 1e-4),
    (dict(weight_planes=False), 1e-4),        # node-level GEMMs on pre-split weight planes (fd_gemm tiles 12-14) vs tiles 2 / 4 / 10
    (dict(defer_node_dw=False), 1e-4),        # the grouped launch at the end of its own block instead of behind the next fused backward
    (dict(fused_embed_bwd=False), 1e-4),
    (dict(zb_from_edge=False), 1e-4),
    (dict(packed_gates=False), 1e-4),
    (dict(edge_dynamic_tiles=False), 1e-4),
    (dict(edge_shape=8), 1e-4),               # (default at these sizes: the 4-wave shape)
    (dict(fused_ln_bwd=False), 1e-4),
    (dict(fused_ln_bwd=False, packed_gates=False), 1e-4),
    (dict(fused_edge=False, fused_embed=False), 2e-4),
]


def _group(key):
    """the first comparison group that switches `key` alone"""
    return next(g for g in GROUPS if set(g[0]) == {key})


def _compare(dev, B, N, blocks, groups=GROUPS):
    l1, g1 = _step(dev, B, N, blocks)
    for off, tol in groups:
        l0, g0 = _step(dev, B, N, blocks, **off)
        assert abs(l0 - l1) <= 2e-6 * abs(l0), (off, l0, l1)
        assert set(g0) == set(g1)
        errs = {n: float((g1[n] - g0[n]).abs().max() / (g0[n].abs().max() + 1e-3)) for n in g0 if not n.endswith("linear_b.bias")}
        for n in g0:
            if n.endswith("linear_b.bias"):
                assert float(g1[n].abs().max()) < 1e-4 and float(g0[n].abs().max()) < 1e-4
        worst = max(errs, key=errs.get)
        assert errs[worst] < tol, (off, worst, errs[worst])


def test_switches_emu(use_emu):
    # (the heads-per-block shapes of the IPA kernel and the sequence-attention backward have kernel-level interpreter tests of
    #  their own -- tests/test_ipa_flash.py -- and run inside a step on the GPU tier)
    _compare("cpu", B=2, N=8, blocks=1, groups=[g for g in GROUPS if not ({"flash_ipa_hpb"} & set(g[0]))])


def test_switches_two_blocks_emu(use_emu):
    # with an edge transition between the blocks: the fused LayerNorm-backward / dzb W40 prologue against the separate kernels
    _compare("cpu", B=1, N=8, blocks=2, groups=[_group("flash_ipa_bwd"), _group("packed_gates"), _group("edge_dynamic_tiles"),
                                                _group("edge_shape"), _group("fused_ln_bwd"), _group("defer_node_dw")])


def _dynamic_vs_static(dev, B, N, blocks):
    """the dynamic tile hand-out of the fused edge kernels inside a full step: with two persistent blocks per launch (edge_blocks=2)
    every fused edge / embedder-backward launch of these sizes has >= 4 tiles per block and takes the atomic-counter path; the
    same step with the static stride must give the same gradients (the default-size comparison was vacuous)."""
    from se3_diffusion_amd import ops
    n0 = ops.STATS["edge_dynamic_launches"]
    l1, g1 = _step(dev, B, N, blocks, edge_blocks=2)
    assert ops.STATS["edge_dynamic_launches"] > n0, "the dynamic path was not taken"
    n1 = ops.STATS["edge_dynamic_launches"]
    l0, g0 = _step(dev, B, N, blocks, edge_blocks=2, edge_dynamic_tiles=False)
    assert ops.STATS["edge_dynamic_launches"] == n1
    assert abs(l0 - l1) <= 2e-6 * abs(l0)
    for n in g0:
        if n.endswith("linear_b.bias"):
            continue
        err = float((g1[n] - g0[n]).abs().max() / (g0[n].abs().max() + 1e-3))
        assert err < 1e-4, (n, err)


def test_dynamic_tiles_emu(use_emu):
    _dynamic_vs_static("cpu", B=2, N=16, blocks=2)


@pytest.mark.gpu
def test_dynamic_tiles_gpu(hip_lib):
    _dynamic_vs_static("cuda", B=2, N=32, blocks=2)


def test_options_override_restores():
    was = options.opts.fused_edge
    with options.override(fused_edge=not was):
        assert options.opts.fused_edge != was
    assert options.opts.fused_edge == was
    with pytest.raises(AttributeError):
        with options.override(no_such_option=1):
            pass


@pytest.mark.gpu
def test_switches_gpu(hip_lib):
    _compare("cuda", B=2, N=24, blocks=2)
    # (B=4 x N=128, one block: everything but the unfused-edge groups, which need an edge transition)
    _compare("cuda", B=4, N=128, blocks=1, groups=[g for g in GROUPS if not ({"grouped_pair_dw", "fused_edge"} & set(g[0]))])
