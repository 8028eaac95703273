"""LayerNorm folded into the Linear that consumes it (csrc/fd_ln_gemm.hip) against float64: the same outputs as
fd_layernorm_fwd followed by fd_gemm (norm1 -> linear1 / norm2 -> in_proj | post_tfmr of the sequence transformer,
model/ipa_pytorch.py:584-595,638).

Tolerance: exact fp32 products, fp32 sums -- 5e-6 of the tensor maximum for the normalised rows, 1e-5 for the product."""
import pytest
import torch

from se3_diffusion_amd import ops

mv = ops.mv


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def _run(dev, M, N, K, seed=0, relu=False, resid=False, scale=False, ln_out=True, mean_shift=0.0, ln_cols=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    x = (rn(M, K) * 1.5 + mean_shift).to(dev)
    W, b = rn(N, K, sc=0.08).to(dev), rn(N, sc=0.3).to(dev)
    Kn = ln_cols or K
    gamma, beta = (1 + rn(Kn, sc=0.2)).to(dev), rn(Kn, sc=0.2).to(dev)
    res = rn(M, N).to(dev) if resid else None
    rs = (torch.rand(M, generator=g) > 0.3).float().to(dev) if scale else None
    out = torch.full((M, N), float("nan"), device=dev)
    y = torch.full((M, K), float("nan"), device=dev) if ln_out else None
    assert ops.ln_linear_ok(mv(x), mv(W), M, N, K)
    ops.ln_linear(mv(x), gamma, beta, mv(W), b, mv(out), M, N, K, relu=relu, resid=None if res is None else mv(res),
                  ln_rowscale=rs, ln_out=None if y is None else mv(y), ln_cols=ln_cols)
    xd = x.double().cpu()
    xn = xd[:, :Kn]
    mean = xn.mean(-1, keepdim=True)
    var = ((xn - mean) ** 2).mean(-1, keepdim=True)
    yr = (xn - mean) / torch.sqrt(var + 1e-5) * gamma.double().cpu() + beta.double().cpu()
    if rs is not None:
        yr = yr * rs.double().cpu()[:, None]
    yr = torch.cat([yr, xd[:, Kn:]], 1)            # (ln_cols: the columns behind it pass through)
    ref = yr @ W.double().cpu().T + b.double().cpu()
    if relu:
        ref = torch.relu(ref)
    if res is not None:
        ref = ref + res.double().cpu()
    if y is not None:
        assert rel(y, yr) < 5e-6, rel(y, yr)
    assert rel(out, ref) < 1e-5, rel(out, ref)
    if K % 64 or ln_cols:
        return                 # (fd_layernorm_fwd takes multiples of 64 only)
    # against the two launches it replaces
    y2, o2 = torch.empty(M, K, device=dev), torch.empty(M, N, device=dev)
    ops.layernorm(mv(x), gamma, beta, mv(y2), M, K, rowscale=rs)
    ops.linear(mv(y2), mv(W), b, mv(o2), M, N, K, relu=relu, resid=None if res is None else mv(res))
    assert rel(out, o2.double().cpu()) < 1e-5


def test_ln_gemm_emu(use_emu):
    _run("cpu", 40, 72, 320, relu=True)                       # ragged row / column tiles, 10 groups per wave
    _run("cpu", 33, 40, 256, seed=1, resid=True, scale=True)
    _run("cpu", 32, 32, 64, seed=2, ln_out=False)             # waves 2, 3 hold one group, ...
    _run("cpu", 8, 8, 8, seed=3)                              # ... or none
    _run("cpu", 40, 72, 320, seed=4, ln_cols=256)             # [LayerNorm(x[:, :256]) | x[:, 256:]] (ipa_ln + skip concat)


@pytest.mark.gpu
def test_ln_gemm_gpu(hip_lib):
    _run("cuda", 128, 320, 320, relu=True)
    _run("cuda", 128, 960, 320, seed=1, scale=True)
    _run("cuda", 256, 256, 320, seed=2, resid=True)
    _run("cuda", 1000, 320, 320, seed=3, resid=True, relu=True, mean_shift=3.0)
    _run("cuda", 50, 72, 200, seed=4, ln_out=False)
    _run("cuda", 128, 960, 320, seed=5, ln_cols=256)


def _infer(dev, B, N, blocks, **kw):
    from oracle import framediff_oracle as fo
    from se3_diffusion_amd import options, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    with options.override(**{k: v for k, v in kw.items() if k != "static_cache"}):
        m = ScoreNetwork(ts.base_model_conf(blocks), diffuser=None)
        m.load_state_dict(fo.synth_params(seed=21, conf=dict(fo.CONF, num_blocks=blocks)), strict=True)
        m = m.to(dev).eval()
        batch = ts.synthetic_batch(B, N, dev, seed=9)
        if kw.pop("static_cache", False):
            m._fd_static = {}           # as sampler.sample does: weight-derived constants are built once and reused
        with torch.no_grad():
            out = m(batch)
            if hasattr(m, "_fd_static"):
                out = m(batch)          # the second forward runs entirely from the cache
        return {k: out[k].double().cpu() for k in ("rot_score", "trans_score", "psi", "rigids")}


def _fold_vs_launches(dev, B, N, blocks):
    a, b = _infer(dev, B, N, blocks), _infer(dev, B, N, blocks, ln_fold=False)
    for k in a:
        assert float((a[k] - b[k]).abs().max()) <= 2e-5 * float(b[k].abs().max() + 1e-3), k
    # the sampler's configuration (static cache: folded per-residue terms, the skip_embed products of all blocks as one GEMM into
    # the column ranges of one [R, 320 nb] buffer) against the same with one skip_embed launch per block, and against `b`
    c = _infer(dev, B, N, blocks, static_cache=True)
    d = _infer(dev, B, N, blocks, static_cache=True, merge_skip_embed=False)
    for k in a:
        assert float((c[k] - d[k]).abs().max()) <= 2e-6 * float(d[k].abs().max() + 1e-3), k
        assert float((c[k] - b[k]).abs().max()) <= 2e-5 * float(b[k].abs().max() + 1e-3), k
    # the edge transitions of a lone backbone (<= 16,384 pair rows) run on the column-split kernel (options.edge_pair,
    # csrc/fd_edge_mlp_pair.hip): the network's outputs against the 4-wave shape it replaces
    e = _infer(dev, B, N, blocks, static_cache=True, edge_pair=False)
    for k in a:
        assert float((c[k] - e[k]).abs().max()) <= 2e-5 * float(e[k].abs().max() + 1e-3), k
    # the embedders' first layers on per-residue features padded to K = 72 (options.embed_first_padded: fd_node_feats_ld, p | q of the
    # edge embedder as one product read through FdEdgeEmbedDesc.ld_pq) against the K = 65 / 33 launches
    e = _infer(dev, B, N, blocks, static_cache=True, embed_first_padded=False)
    for k in a:
        assert float((c[k] - e[k]).abs().max()) <= 2e-5 * float(e[k].abs().max() + 1e-3), k


def test_inference_fold_vs_layernorm_launches_emu(use_emu):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _fold_vs_launches("cpu", 1, 10, 2)


@pytest.mark.gpu
def test_inference_fold_vs_layernorm_launches_gpu(hip_lib):
    _fold_vs_launches("cuda", 1, 128, 4)
    _fold_vs_launches("cuda", 3, 50, 2)
