"""Alternative launch sequences of the same arithmetic give the same training step: IPA's four projections of s as ONE
GEMM over back-to-back weights (network._PROJ_MERGE; optim.FlatAdam lays them out so) and the sequence-transformer
attention in one launch (network.FUSED_SEQ_ATTN) against the separate launches -- model/ipa_pytorch.py:340-374,584-593.

Tolerance: 5e-5 of (each gradient's maximum + 1e-3) -- the merged GEMM has other tile shapes, the fused attention another
summation order, both fp32-accurate (measured 2.1e-5 at B=4 x N=128, the size of either path's distance to the oracle).  linear_b.bias is left out: its gradient is analytically zero (a softmax is invariant
to a shift of its logits) and numerically the round-off of a sum of O(1) terms (|g| ~ 5e-8 at either setting)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import framediff_oracle as fo  # noqa: E402
from se3_diffusion_amd import network as nw, train_step as ts  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402
from se3_diffusion_amd.optim import FlatAdam  # noqa: E402


def _step(dev, B, N, blocks, merge, seq_fused, embed_dw="0"):
    was = nw._PROJ_MERGE, nw.FUSED_SEQ_ATTN, nw.SEQ_ATTN_MIN_ROWS
    was_e = nw._EMBED_DW_GROUPED, nw._EMBED_DW_DIAG, nw._EMBED_DW_MIN_ROWS, nw._EMBED_DW_BLOCKS
    nw._PROJ_MERGE, nw.FUSED_SEQ_ATTN, nw.SEQ_ATTN_MIN_ROWS = merge, seq_fused, 0      # (0: the fused kernel at any size)
    nw._EMBED_DW_GROUPED, nw._EMBED_DW_DIAG, nw._EMBED_DW_MIN_ROWS = embed_dw == "1", embed_dw == "diag", 0
    nw._EMBED_DW_BLOCKS = 8 if embed_dw != "0" else nw._EMBED_DW_BLOCKS
    try:
        conf = dict(fo.CONF, num_blocks=blocks)
        m = ScoreNetwork(ts.base_model_conf(blocks), diffuser=None)
        m.load_state_dict(fo.synth_params(seed=11, conf=conf), strict=True)
        m = m.to(dev).train()
        opt = FlatAdam(m.parameters(), lr=1e-4, adjacent=m.flat_layout_groups())
        ipa = m.score_model.trunk["ipa_0"]
        assert ipa.linear_q.weight.data_ptr() + 4 * ipa.linear_q.weight.numel() == ipa.linear_kv.weight.data_ptr()
        assert (nw._proj_views(dict(m.named_parameters()), "score_model.trunk.ipa_0") is not None) == merge
        batch = ts.synthetic_batch(B, N, dev, seed=5)
        batch["t"][0] = 0.1
        cb = {k: v.cpu() for k, v in batch.items()}
        gt37, _ = fo.backbone_atoms(cb["rigids_0"][..., :4], cb["rigids_0"][..., 4:], cb["torsion_angles_sin_cos"][..., 2, :])
        opt.zero_grad()
        loss = ts.dsm_loss(batch, m(batch), gt37.to(dev))
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().double().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        nw._PROJ_MERGE, nw.FUSED_SEQ_ATTN, nw.SEQ_ATTN_MIN_ROWS = was
        nw._EMBED_DW_GROUPED, nw._EMBED_DW_DIAG, nw._EMBED_DW_MIN_ROWS, nw._EMBED_DW_BLOCKS = was_e


def _compare(dev, B, N, blocks, embed_dw="0"):
    l0, g0 = _step(dev, B, N, blocks, False, False)
    l1, g1 = _step(dev, B, N, blocks, True, True, embed_dw)
    assert abs(l0 - l1) <= 2e-6 * abs(l0), (l0, l1)
    assert set(g0) == set(g1)
    worst = max(float((g1[n] - g0[n]).abs().max() / (g0[n].abs().max() + 1e-3)) for n in g0 if not n.endswith("linear_b.bias"))
    for n in g0:
        if n.endswith("linear_b.bias"):
            assert float(g1[n].abs().max()) < 1e-4 and float(g0[n].abs().max()) < 1e-4
    assert worst < 5e-5, worst


def test_switches_emu(use_emu):
    _compare("cpu", B=2, N=8, blocks=1)
    # the edge embedder's weight gradients through fd_pair_dw (128 x 128 items) / fd_pair_dw_diag (one pass): both opt-in
    _compare("cpu", B=2, N=8, blocks=1, embed_dw="1")
    _compare("cpu", B=2, N=8, blocks=1, embed_dw="diag")


@pytest.mark.gpu
def test_switches_gpu(hip_lib):
    _compare("cuda", B=2, N=24, blocks=2)
    _compare("cuda", B=4, N=128, blocks=1)
