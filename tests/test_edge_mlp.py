"""Fused edge transition (csrc/fd_edge_mlp.hip) against a float64 restatement of EdgeTransition.forward
(model/ipa_pytorch.py:218-233, split into the z / e_i / e_j parts as trunk.edge_transition_fwd does) and of its dX chain.

Tolerance: split-bf16 arithmetic is fp32-accurate -- 5e-6 of the tensor maximum for the hidden activations and the
pre-LayerNorm output, 2e-5 for the LayerNorm output."""
import numpy as np
import pytest
import torch

from se3_diffusion_amd import ops, options


def _case(dev, B, N, seed):
    g = torch.Generator().manual_seed(seed)
    R, P = B * N, B * N * N
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    t = dict(z=rn(P, 128), P1=rn(R, 384, sc=0.5), Q1=rn(R, 384, sc=0.5), b2=rn(384, sc=0.3), Pf=rn(R, 128, sc=0.5),
             Qf=rn(R, 128, sc=0.5), W1=rn(384, 384, sc=0.08), W2=rn(384, 384, sc=0.08), Wf=rn(128, 384, sc=0.08),
             gamma=1 + rn(128, sc=0.2), beta=rn(128, sc=0.2), emask=(torch.rand(P, generator=g) > 0.2).float(),
             dy=rn(P, 128))
    return {k: v.to(dev) for k, v in t.items()}


def _ref_fwd(t, B, N):
    d = {k: v.double().cpu() for k, v in t.items()}
    R = B * N
    qi = torch.arange(B * N * N) // N
    qj = (qi // N) * N + torch.arange(B * N * N) % N
    h1 = torch.relu(d["z"] @ d["W1"][:, :128].T + d["P1"][qi] + d["Q1"][qj])
    h2 = torch.relu(h1 @ d["W2"].T + d["b2"])
    # final_layer(trunk(x) + x), ipa_pytorch.py:231: the z part of x rides on the first 128 hidden units
    y = h2 @ d["Wf"].T + d["z"] @ d["Wf"][:, :128].T + d["Pf"][qi] + d["Qf"][qj]
    mean = y.mean(-1, keepdim=True)
    var = ((y - mean) ** 2).mean(-1, keepdim=True)
    out = ((y - mean) / torch.sqrt(var + 1e-5) * d["gamma"] + d["beta"]) * d["emask"][:, None]
    return h1, h2, y, out, mean[:, 0], 1 / torch.sqrt(var + 1e-5)[:, 0]


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def _run(dev, B, N, seed=0, blocks=0):
    # (the checks below compare the inference forward with the training forward bit for bit: the 16-rows-per-wave shapes; the
    #  column-split kernel small inference launches take by default has its own test, _pair)
    with options.override(edge_pair=False):
        _run_16(dev, B, N, seed, blocks)


def _run_16(dev, B, N, seed=0, blocks=0):
    t = _case(dev, B, N, seed)
    P = B * N * N
    img = ops.edge_mlp_pack(t["W1"], t["W2"], t["Wf"])
    e = lambda *s: torch.empty(*s, device=dev)
    out, h1, h2z, y, mean, rstd = e(P, 128), e(P, 384), e(P, 384), e(P, 128), e(P), e(P)
    mh1 = torch.zeros(P, 12, dtype=torch.int32, device=dev); mh2 = torch.zeros(P, 12, dtype=torch.int32, device=dev)
    # training outputs: h1, h2 + [z | 0 | 0] (the operand of the final layer's weight gradient), the packed signs of h1 / h2
    ops.edge_mlp(t["z"], img, out, P, N, p1=t["P1"], q1=t["Q1"], bias2=t["b2"], pf=t["Pf"], qf=t["Qf"], gamma=t["gamma"],
                 beta=t["beta"], rowscale=t["emask"], save1=h1, save2=h2z, y=y, mean=mean, rstd=rstd, blocks=blocks,
                 mask1=mh1, mask2=mh2)
    rh1, rh2, ry, rout, rmean, rrstd = _ref_fwd(t, B, N)
    rh2z = rh2.clone()
    rh2z[:, :128] += t["z"].double().cpu()
    assert rel(h1, rh1) < 5e-6 and rel(h2z, rh2z) < 5e-6 and rel(y, ry) < 5e-6
    assert rel(out, rout) < 2e-5 and rel(mean, rmean) < 5e-6 and rel(rstd, rrstd) < 2e-5
    # without the optional outputs (sampling)
    out2 = e(P, 128)
    ops.edge_mlp(t["z"], img, out2, P, N, p1=t["P1"], q1=t["Q1"], bias2=t["b2"], pf=t["Pf"], qf=t["Qf"], gamma=t["gamma"],
                 beta=t["beta"], rowscale=t["emask"], blocks=blocks)
    assert torch.equal(out, out2)
    # with the fourth layer: zb = W40 out + b40 (the next IPA block's linear_b / down_z, ipa_pytorch.py:380-386,455), with
    # and without the training saves; the other outputs are unchanged
    gz = torch.Generator().manual_seed(seed + 100)
    W40, b40 = (torch.randn(40, 128, generator=gz) * 0.1).to(dev), torch.randn(40, generator=gz).to(dev)
    img4 = ops.edge_mlp_pack(t["W1"], t["W2"], t["Wf"], W40=W40)
    for saves in (False, True):
        out3, zb = e(P, 128), torch.full((P, 40), float("nan"), device=dev)
        kw = dict(save1=e(P, 384), save2=e(P, 384), y=e(P, 128), mean=e(P), rstd=e(P), mask1=torch.zeros_like(mh1),
                  mask2=torch.zeros_like(mh2)) if saves else {}
        ops.edge_mlp(t["z"], img4, out3, P, N, p1=t["P1"], q1=t["Q1"], bias2=t["b2"], pf=t["Pf"], qf=t["Qf"], gamma=t["gamma"],
                     beta=t["beta"], rowscale=t["emask"], blocks=blocks, zb_out=zb, zb_bias=b40, **kw)
        assert torch.equal(out, out3)
        rzb = rout @ W40.double().cpu().T + b40.double().cpu()
        assert rel(zb, rzb) < 2e-5, rel(zb, rzb)
    # packed ReLU gates: the forward's sign masks (bit 4 nb + e of word (row, chunk c, g) <-> unit 128 c + 16 nb + 4 g + e)
    unit = torch.arange(384)
    c, nb, g, ee = unit // 128, (unit % 128) // 16, (unit % 16) // 4, unit % 4
    h2_own = h2z.cpu().clone()
    h2_own[:, :128] -= t["z"].cpu()
    for mh, h in ((mh1, h1.cpu() > 0), (mh2, None)):
        words = mh.cpu().long() & 0xFFFFFFFF
        bits = ((words[:, (4 * c + g)] >> (4 * nb + ee)) & 1).bool()
        if h is None:
            # h2 alone is not an output; h2z - z recovers it up to the rounding of the add, so only clearly non-zero units decide
            clear = h2_own.abs() > 1e-4
            assert torch.equal(bits[clear], (h2_own > 0)[clear]) and float(((bits != (rh2 > 0)).float().mean())) < 1e-4
        else:
            assert torch.equal(bits, h)
    gate2 = ((mh2.cpu().long() & 0xFFFFFFFF)[:, (4 * c + g)] >> (4 * nb + ee)) & 1
    # backward chain: u = dy Wf ; d2 = [h2 > 0] u ; d1 = [h1 > 0] d2 W2 ; dz = u[:, :128] + d1 W1[:, :128]
    imgT = ops.edge_mlp_pack(t["W1"], t["W2"], t["Wf"], backward=True)
    dz, d2, d1 = e(P, 128), e(P, 384), e(P, 384)
    ops.edge_mlp(t["dy"], imgT, dz, P, N, gmask1=mh2, gmask2=mh1, save1=d2, save2=d1, backward=True, blocks=blocks)
    dd = {k: v.double().cpu() for k, v in t.items()}
    # (gates are taken from the kernel's own masks: a unit whose fp64 pre-activation is within round-off of zero may flip)
    g2, g1 = gate2.bool(), h1.cpu() > 0
    flip = float((g2 != (rh2 > 0)).float().mean() + (g1 != (rh1 > 0)).float().mean())
    assert flip < 1e-4
    rd2 = (dd["dy"] @ dd["Wf"]) * g2
    rd1 = (rd2 @ dd["W2"]) * g1
    rdz = dd["dy"] @ dd["Wf"][:, :128] + rd1 @ dd["W1"][:, :128]
    assert rel(d2, rd2) < 5e-6 and rel(d1, rd1) < 5e-6 and rel(dz, rdz) < 5e-6
    # fused prologue: the kernel's input dy = LayerNorm backward of the upstream gradient (x emask), with and without the IPA
    # pair-projection term dzb W40 (of the block behind the transition) added to the upstream gradient first -- against the
    # unfused pieces: fd_layernorm_bwd (+ a float64 dzb W40) and the plain gated backward
    gu = torch.Generator().manual_seed(seed + 200)
    up = torch.randn(P, 128, generator=gu).to(dev)
    dzb = (torch.randn(P, 40, generator=gu) * 0.5).to(dev)
    mvw = lambda a: (a, 0, a.shape[-1])
    for with_zb, with_up in ((False, True), (True, True), (True, False)):
        up_full = (up.double() if with_up else 0) + (dzb.double() @ W40.double() if with_zb else 0)
        up32 = up_full.float().contiguous()
        dyr, dgr, dbr = e(P, 128), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        ops.layernorm_bwd(mvw(up32), mvw(y), t["gamma"], mean, rstd, mvw(dyr), P, 128, rowscale=t["emask"], dgamma=dgr, dbeta=dbr)
        dzr, d2r, d1r = e(P, 128), e(P, 384), e(P, 384)
        ops.edge_mlp(dyr, imgT, dzr, P, N, gmask1=mh2, gmask2=mh1, save1=d2r, save2=d1r, backward=True, blocks=blocks)
        imgB = ops.edge_mlp_pack_bwd(t["Wf"], t["W2"], t["W1"], W40=W40 if with_zb else None)
        dyf, dzf, d2f, d1f = e(P, 128), e(P, 128), e(P, 384), e(P, 384)
        dgf, dbf = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        ops.edge_mlp(up if with_up else None, imgB, dzf, P, N, gmask1=mh2, gmask2=mh1, save1=d2f, save2=d1f, backward=True,
                     blocks=blocks, ln_y=y, ln_mean=mean, ln_rstd=rstd, ln_gamma=t["gamma"], ln_rowscale=t["emask"], dy_out=dyf,
                     ln_dgamma=dgf, ln_dbeta=dbf, dzb=dzb if with_zb else None)
        ref = lambda a: a.double().cpu()
        assert rel(dyf, ref(dyr)) < 1e-5, (with_zb, with_up, rel(dyf, ref(dyr)))
        assert rel(dgf, ref(dgr)) < 1e-5 and rel(dbf, ref(dbr)) < 1e-5
        assert rel(d2f, ref(d2r)) < 1e-5 and rel(d1f, ref(d1r)) < 1e-5 and rel(dzf, ref(dzr)) < 1e-5


def _same_in_both_shapes(dev, B, N, seed, blocks):
    """the two shapes of the kernel (4 waves x 64-row tiles on two blocks per CU / 8 waves x 128-row tiles on one) run the same
    arithmetic in the same order per pair row: forward outputs, saves, packed masks and the zb layer bit for bit"""
    t = _case(dev, B, N, seed)
    P = B * N * N
    gz = torch.Generator().manual_seed(seed + 100)
    W40, b40 = (torch.randn(40, 128, generator=gz) * 0.1).to(dev), torch.randn(40, generator=gz).to(dev)
    img = ops.edge_mlp_pack(t["W1"], t["W2"], t["Wf"], W40=W40)
    res = []
    for shape in (4, 8):
        e = lambda *s: torch.full(s, float("nan"), device=dev)
        o = dict(out=e(P, 128), h1=e(P, 384), h2=e(P, 384), y=e(P, 128), mean=e(P), rstd=e(P), zb=e(P, 40),
                 m1=torch.zeros(P, 12, dtype=torch.int32, device=dev), m2=torch.zeros(P, 12, dtype=torch.int32, device=dev))
        with options.override(edge_shape=shape):
            ops.edge_mlp(t["z"], img, o["out"], P, N, p1=t["P1"], q1=t["Q1"], bias2=t["b2"], pf=t["Pf"], qf=t["Qf"],
                         gamma=t["gamma"], beta=t["beta"], rowscale=t["emask"], save1=o["h1"], save2=o["h2"], y=o["y"],
                         mean=o["mean"], rstd=o["rstd"], blocks=blocks, zb_out=o["zb"], zb_bias=b40, mask1=o["m1"], mask2=o["m2"])
        res.append(o)
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def _pair(dev, B, N, seed, blocks=0):
    """the column-split kernel (csrc/fd_edge_mlp_pair.hip: two waves per 16-row group, inference forward): z' and zb against the
    float64 restatement (the tolerances of the 16-row kernels) and against the 4-wave shape (same products, the halves of layer 3 /
    LayerNorm / zb added in another order: fp32 rounding); what the size rule picks, and edge_pair=False"""
    t = _case(dev, B, N, seed)
    P = B * N * N
    gz = torch.Generator().manual_seed(seed + 100)
    W40, b40 = (torch.randn(40, 128, generator=gz) * 0.1).to(dev), torch.randn(40, generator=gz).to(dev)
    imgs = {False: ops.edge_mlp_pack(t["W1"], t["W2"], t["Wf"]), True: ops.edge_mlp_pack(t["W1"], t["W2"], t["Wf"], W40=W40)}
    rout = _ref_fwd(t, B, N)[3]
    rzb = rout @ W40.double().cpu().T + b40.double().cpu()

    def go(zbv, **ov):
        out, zb = torch.full((P, 128), float("nan"), device=dev), torch.full((P, 40), float("nan"), device=dev)
        with options.override(**ov):
            ops.edge_mlp(t["z"], imgs[zbv], out, P, N, p1=t["P1"], q1=t["Q1"], bias2=t["b2"], pf=t["Pf"], qf=t["Qf"],
                         gamma=t["gamma"], beta=t["beta"], rowscale=t["emask"], blocks=blocks,
                         **(dict(zb_out=zb, zb_bias=b40) if zbv else {}))
        return out, zb

    for zbv in (False, True):
        o2, z2 = go(zbv, edge_shape=2)
        o4, z4 = go(zbv, edge_shape=4)
        assert rel(o2, rout) < 2e-5, rel(o2, rout)
        assert rel(o2, o4.double().cpu()) < 3e-6, rel(o2, o4.double().cpu())
        if zbv:
            assert rel(z2, rzb) < 2e-5 and rel(z2, z4.double().cpu()) < 3e-6
        if P <= 16384:
            oa, za = go(zbv)                      # the size rule: at most one 16-row group per SIMD -> the column-split kernel
            assert torch.equal(oa, o2) and (not zbv or torch.equal(za, z2))
            ob, zbb = go(zbv, edge_pair=False)
            assert torch.equal(ob, o4) and (not zbv or torch.equal(zbb, z4))


def test_edge_mlp_pair_emu(use_emu):
    _pair("cpu", B=1, N=12, seed=11)               # 144 rows: two full 64-row tiles + a ragged one
    _pair("cpu", B=1, N=17, seed=12, blocks=1)     # 289 rows: five tiles walked by one block


@pytest.mark.gpu
def test_edge_mlp_pair_gpu(hip_lib):
    _pair("cuda", B=1, N=12, seed=11)
    _pair("cuda", B=1, N=128, seed=13)             # 16,384 rows: the lone backbone the kernel exists for, one tile per block
    _pair("cuda", B=1, N=67, seed=14, blocks=5)    # ragged tail, several tiles per block
    _pair("cuda", B=2, N=128, seed=15)             # above the size rule's bound: forced, two tiles per block


def test_edge_mlp_emu(use_emu):
    _run("cpu", B=1, N=12)            # 144 rows: one full tile + a ragged one
    _run("cpu", B=1, N=17, seed=4, blocks=1)     # 5 tiles on 1 block (>= 4 per block): the dynamic tile hand-out


def test_edge_mlp_w8_emu(use_emu):
    """the one-block-per-CU shape (csrc/fd_edge_mlp_w8.hip): every check of the default shape, static and dynamic tile order"""
    with options.override(edge_shape=8):
        _run("cpu", B=1, N=13)                      # 169 rows: one full 128-row tile + a ragged one
        _run("cpu", B=1, N=23, seed=5, blocks=1)    # 529 rows: 5 tiles on 1 block: the dynamic hand-out
    _same_in_both_shapes("cpu", B=1, N=14, seed=6, blocks=0)


@pytest.mark.gpu
def test_edge_mlp_gpu(hip_lib):
    _run("cuda", B=1, N=12)
    _run("cuda", B=2, N=50, seed=1)
    _run("cuda", B=3, N=128, seed=2)               # 384 tiles: persistent blocks walk several tiles
    _run("cuda", B=1, N=67, seed=3, blocks=5)      # few blocks, many tiles each, ragged tail


@pytest.mark.gpu
def test_edge_mlp_w8_gpu(hip_lib):
    with options.override(edge_shape=8):
        _run("cuda", B=1, N=12)
        _run("cuda", B=2, N=50, seed=1)
        _run("cuda", B=1, N=67, seed=3, blocks=5)
    _run("cuda", B=9, N=128, seed=2)               # 147,456 rows: the shape the entry point picks by size (1,152 tiles on 256 blocks)
    _same_in_both_shapes("cuda", B=3, N=128, seed=7, blocks=0)
    _same_in_both_shapes("cuda", B=1, N=67, seed=8, blocks=3)
