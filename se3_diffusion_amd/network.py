"""ScoreNetwork forward/backward as a sequence of HIP kernel launches.

Mirrors the reference computation graph (model/score_network.py:170-215,
model/ipa_pytorch.py:611-672) stage by stage; every stage has a hand-written
backward.  Stages exchange plain fp32 tensors; `P` is the parameter dict (state_dict
names), `G` the gradient dict (same names, accumulated into), `sv` a dict of tensors
saved by the forward for the backward.

Stage list per trunk block b (A3 of SURVEY.md):
  ipa        x1 = s + m * IPA(s, z, T)                     ipa_pytorch.py:625-631
  ln_skip    u0 = [LN(x1) | W_skip s_init]                 :632-635
  tfmr x2    post-norm encoder layers on u (320)           :636-637
  post       n2 = u0[:, :256] + W_post u2                  :638
  node_tr    n3 = m * LN(n2 + MLP(n2))                     :639-640
  bb_update  T' = T o (W_bb (n3 * d))                      :641-644
  edge_tr    z' = mm * LN(W_f(relu(W_b relu(W_a x)) + x))  :646-649
"""
from __future__ import annotations

import math

import torch

from . import hip, ops
from .ops import empty, zeros, mv, lib
from .options import opts

H, C, PQ, PV, CZ4 = 8, 256, 8, 12, 32
CS, CZ = 256, 128
LDP = 6816
LDF = 2688
ZB = 40
TD, TH, THD = 320, 4, 80  # transformer width, heads, head dim


def check_conf(conf):
    """Which model_conf keys are free and which are frozen (INTEGRATION.md "model_conf").  Free: ipa.num_blocks,
    ipa.seq_tfmr_num_layers (host loops), ipa.coordinate_scaling, dropout (unused by the reference's forward too).  Frozen at
    config/base.yaml's values, because the kernels bake them into register / LDS layouts: node_embed_size = ipa.c_s = 256,
    edge_embed_size = ipa.c_z = 128, ipa.c_hidden = 256, ipa.c_skip = 64, ipa.no_heads = 8, no_qk_points = 8, no_v_points = 12,
    seq_tfmr_num_heads = 4, embed.index_embed_size = 32, embed.num_bins = 22, embed.embed_self_conditioning = True."""
    ipa = conf.ipa
    want = dict(c_s=256, c_z=128, c_hidden=256, c_skip=64, no_heads=8, no_qk_points=8, no_v_points=12,
                seq_tfmr_num_heads=4)
    for k, v in want.items():
        if getattr(ipa, k) != v:
            raise NotImplementedError(
                f"the gfx950 kernels are built for config/base.yaml dimensions; model.ipa.{k}={getattr(ipa, k)} != {v}")
    if conf.node_embed_size != 256 or conf.edge_embed_size != 128:
        raise NotImplementedError("node_embed_size/edge_embed_size must be 256/128")
    if int(getattr(ipa, "seq_tfmr_num_layers", 2)) < 1:
        raise NotImplementedError(f"model.ipa.seq_tfmr_num_layers={ipa.seq_tfmr_num_layers}: at least one transformer layer per block")
    e = conf.embed
    if e.index_embed_size != 32 or e.num_bins != 22 or not e.embed_self_conditioning:
        raise NotImplementedError("embed config must match config/base.yaml (index 32, 22 bins, self-conditioning)")


# --------------------------------------------------------------------------- param grads
def _lin_grads(G, wname, bname, dy, x, M, N, K, w_off=0, w_ld=None):
    if G is None:
        return
    db = G[bname] if (bname is not None and bname in G) else None
    # node-level calls run on the gradient side stream (ops.side): dy / x must not be written again by the caller
    if wname in G:
        W = G[wname]
        if ops.queue_dw(dy, x, (W, w_off, w_ld if w_ld is not None else W.shape[1]), M, N, K, db=db):
            return          # part of the block's grouped launch (ops.flush_dw in trunk.backward)
        ops.side(lambda: ops.linear_dw(dy, x, (W, w_off, w_ld if w_ld is not None else W.shape[1]), M, N, K, db=db),
                 (dy[0], x[0]), M)
    elif db is not None:
        ops.side(lambda: ops.bias_grad(dy, db, M, N), (dy[0],), M)


# --------------------------------------------------------------------------- embedder
def mlp3_ln_fwd(P, pre, x, M, K0, Cc, rowscale, W0=None):
    """Linear-ReLU-Linear-ReLU-Linear-LayerNorm (* rowscale) -- score_network.py:67-86,194-195.  W0: the first layer's weight
    zero-padded to the K0 columns of a padded x (inference)."""
    dev = x[0]
    h1 = empty((M, Cc), dev); h2 = empty((M, Cc), dev); h3 = empty((M, Cc), dev); y = empty((M, Cc), dev)
    mean = empty((M,), dev); rstd = empty((M,), dev)
    ops.linear(x, mv(P[f"{pre}.0.weight"] if W0 is None else W0), P[f"{pre}.0.bias"], mv(h1), M, Cc, K0, relu=True)
    ops.linear(mv(h1), mv(P[f"{pre}.2.weight"]), P[f"{pre}.2.bias"], mv(h2), M, Cc, Cc, relu=True)
    ops.linear(mv(h2), mv(P[f"{pre}.4.weight"]), P[f"{pre}.4.bias"], mv(h3), M, Cc, Cc)
    ops.layernorm(mv(h3), P[f"{pre}.5.weight"], P[f"{pre}.5.bias"], mv(y), M, Cc, rowscale=rowscale, save=(mean, rstd))
    return y, dict(x=x, h1=h1, h2=h2, h3=h3, mean=mean, rstd=rstd, rowscale=rowscale, M=M, K0=K0, C=Cc)


def mlp3_ln_bwd(P, G, pre, sv, dy):
    M, K0, Cc = sv["M"], sv["K0"], sv["C"]
    dev = dy
    dh3 = empty((M, Cc), dev)
    ops.layernorm_bwd(mv(dy), mv(sv["h3"]), P[f"{pre}.5.weight"], sv["mean"], sv["rstd"], mv(dh3), M, Cc,
                      rowscale=sv["rowscale"], dgamma=G[f"{pre}.5.weight"], dbeta=G[f"{pre}.5.bias"])
    _lin_grads(G, f"{pre}.4.weight", f"{pre}.4.bias", mv(dh3), mv(sv["h2"]), M, Cc, Cc)
    dh2 = empty((M, Cc), dev)
    ops.linear_dx(mv(dh3), mv(P[f"{pre}.4.weight"]), mv(dh2), M, Cc, Cc, gate=mv(sv["h2"]))
    _lin_grads(G, f"{pre}.2.weight", f"{pre}.2.bias", mv(dh2), mv(sv["h1"]), M, Cc, Cc)
    dh1 = empty((M, Cc), dev)   # (dh3 is still being read by its weight gradient on the side stream)
    ops.linear_dx(mv(dh2), mv(P[f"{pre}.2.weight"]), mv(dh1), M, Cc, Cc, gate=mv(sv["h1"]))
    _lin_grads(G, f"{pre}.0.weight", f"{pre}.0.bias", mv(dh1), sv["x"], M, Cc, K0)


def fused_embed():
    """The fused edge-embedder kernel computes in split-bf16 (fp32-accurate): off in exact-fp32 mode."""
    return opts.fused_embed and not lib().exact_f32


def embed_fwd(P, feats, B, N, cache=None, save=True, zb_next=None):
    """Returns (node, edge, saved, zb): zb = W40 edge + b40 of the first IPA block (zb_next = its (W40, b40)) when the fused
    edge embedder forms it from its output in registers, else None."""
    dev = feats["res_mask"]
    mask = feats["res_mask"]
    tfreq, idenom, lower, upper = ops.feature_tables(dev.device)
    tscaled = feats["tscaled"] if "tscaled" in feats else (feats["t"] * 10000).float().contiguous()   # score_network.py:38,43
    fixed = feats["fixed_mask"]
    seq = feats["seq_idx"]
    R, Pn = B * N, B * N * N
    pre = "embedding_layer.edge_embedder"
    # sampling (static weights, fused edge embedder, opts.embed_first_padded): the three first-layer products of the per-residue
    # features -- node embedder 65 -> 256, the edge embedder's p and q halves 33 -> 128 each -- are K = 65 / 33 launches on the
    # unaligned fp32 tile (9.5 us each on a lone backbone).  With the features written at a row stride of 72 (zero columns behind
    # the 65) and the weights zero-padded once per trajectory they are two launches of the latency GEMM: the node layer, and p | q
    # as the two column halves of ONE [R, 256] product that the fused edge kernel reads through its row stride.
    padded = cache is not None and not save and opts.embed_first_padded and fused_embed()
    NFL = 72 if padded else 65
    nf = empty((R, NFL), dev)
    if padded:
        lib().call("fd_node_feats_ld", seq, tscaled, fixed, tfreq, idenom, nf, NFL, B, N)
        if "embed_first" not in cache:
            Wn, W0e = P["embedding_layer.node_embedder.0.weight"], P[f"{pre}.0.weight"]
            Wn72 = torch.zeros((CS, NFL), device=Wn.device); Wn72[:, :65] = Wn
            Wpq = torch.zeros((2 * CZ, NFL), device=Wn.device); Wpq[:CZ, :33] = W0e[:, :33]; Wpq[CZ:, :33] = W0e[:, 33:66]
            bpq = torch.cat([P[f"{pre}.0.bias"], torch.zeros_like(P[f"{pre}.0.bias"])]).contiguous()
            cache["embed_first"] = (Wn72, Wpq, bpq)
        Wn72, Wpq, bpq = cache["embed_first"]
        node, sv_n = mlp3_ln_fwd(P, "embedding_layer.node_embedder", mv(nf), R, NFL, CS, mask, W0=Wn72)
    else:
        lib().call("fd_node_feats", seq, tscaled, fixed, tfreq, idenom, nf, B, N)
        node, sv_n = mlp3_ln_fwd(P, "embedding_layer.node_embedder", mv(nf), R, 65, CS, mask)
    emask = pair_mask(mask, B, N) if cache is None else cache.setdefault("emask", None)
    if emask is None:
        emask = cache["emask"] = pair_mask(mask, B, N)
    if not fused_embed():
        ef = empty((Pn, 120), dev)
        lib().call("fd_edge_feats", seq, tscaled, fixed, feats["sc_ca_t"], tfreq, idenom, lower, upper, ef, B, N)
        edge, sv_e = mlp3_ln_fwd(P, pre, mv(ef), Pn, 120, CZ, emask)
        return node, edge, dict(node=sv_n, edge=sv_e, emask=emask), None
    # fused: no [P,120] feature tensor, no hidden activations in HBM.  The residue-only part of the first layer
    # (t-embedding + fixed flag of i and of j = the first 33 columns of the node feature) is node-level:
    # p = W0[:, 0:33] pt + b0, q = W0[:, 33:66] pt
    W0 = P[f"{pre}.0.weight"]
    ld_pq = 0
    if padded:
        pq = empty((R, 2 * CZ), dev)
        ops.linear(mv(nf), mv(Wpq), bpq, mv(pq), R, 2 * CZ, NFL)
        p_, q_, ld_pq = pq, pq[:, CZ:], 2 * CZ
    else:
        p_ = empty((R, CZ), dev); q_ = empty((R, CZ), dev)
        ops.linear((nf, 0, 65), (W0, 0, 120), P[f"{pre}.0.bias"], mv(p_), R, CZ, 33)
        ops.linear((nf, 0, 65), (W0, 33, 120), None, mv(q_), R, CZ, 33)
    use_zb = (opts.zb_from_edge and zb_next is not None and zb_next[0].is_contiguous() and zb_next[0].data_ptr() % 16 == 0
              and zb_next[1].data_ptr() % 16 == 0)
    key = ("ee_img", pre, use_zb)
    if cache is not None and key in cache:
        img = cache[key]
    else:
        img = ops.edge_embed_pack(W0, P[f"{pre}.2.weight"], P[f"{pre}.4.weight"], W40=zb_next[0] if use_zb else None)
        if cache is not None:
            cache[key] = img
    edge = empty((Pn, CZ), dev)
    kw = {}
    zb = None
    if use_zb:
        zb = empty((Pn, ZB), dev)
        kw.update(zb_out=zb, zb_bias=zb_next[1])
    if save:
        h1 = empty((Pn, CZ), dev); h2 = empty((Pn, CZ), dev); h3 = empty((Pn, CZ), dev)
        mean = empty((Pn,), dev); rstd = empty((Pn,), dev)
        kw.update(h1=h1, h2=h2, h3=h3, mean=mean, rstd=rstd)
        if opts.packed_gates:
            mh1 = empty((Pn, 4), dev, torch.int32); mh2 = empty((Pn, 4), dev, torch.int32)
            kw.update(mask1=mh1, mask2=mh2)
    ops.edge_embed(seq, feats["sc_ca_t"], idenom, lower, upper, img, p_, q_, P[f"{pre}.2.bias"], P[f"{pre}.4.bias"],
                   P[f"{pre}.5.weight"], P[f"{pre}.5.bias"], edge, Pn, N, rowscale=emask, ld_pq=ld_pq, **kw)
    sv_e = None
    if save:
        # the backward is the unfused MLP backward; its first-layer weight gradient needs the [P,120] feature tensor,
        # which is regenerated there (embed_bwd) instead of being kept alive through the whole step
        sv_e = dict(x=None, h1=h1, h2=h2, h3=h3, mean=mean, rstd=rstd, rowscale=emask, M=Pn, K0=120, C=CZ,
                    mh1=kw.get("mask1"), mh2=kw.get("mask2"),
                    regen=(seq, tscaled, fixed, feats["sc_ca_t"], B, N))
    return node, edge, dict(node=sv_n, edge=sv_e, emask=emask), zb


def embed_regen_early(sv, G):
    """The [P,120] pair features the first-layer weight gradient of the fused edge embedder needs are a function of the
    inputs alone: regenerate them at the START of the backward pass on the gradient side stream (idle then) instead of on
    the main stream at its end, where nothing is left to hide the 146 us behind."""
    se = sv["edge"]
    if G is None or se is None or se.get("x") is not None or "regen" not in se:
        return
    seq, tscaled, fixed, sc, B, N = se["regen"]
    Pn = B * N * N
    if not ops.side_active(sc, Pn):
        return            # (its only reader, the weight gradient, would run on the main stream)
    tfreq, idenom, lower, upper = ops.feature_tables(sc.device)
    ef = empty((Pn, 120), sc)
    ops.side(lambda: lib().call("fd_edge_feats", seq, tscaled, fixed, sc, tfreq, idenom, lower, upper, ef, B, N), (ef,), Pn)
    se["x_early"] = ef


def embed_bwd(P, G, sv, dnode, dedge):
    mlp3_ln_bwd(P, G, "embedding_layer.node_embedder", sv["node"], dnode)
    se = sv["edge"]
    if se.get("x_early") is not None:
        se = dict(se, x=mv(se.pop("x_early")))     # (read by the side stream only: mlp3_ln_bwd's last _lin_grads)
    if se.get("x") is None:
        seq, tscaled, fixed, sc, B, N = se["regen"]
        tfreq, idenom, lower, upper = ops.feature_tables(dedge.device)
        ef = empty((B * N * N, 120), dedge)
        lib().call("fd_edge_feats", seq, tscaled, fixed, sc, tfreq, idenom, lower, upper, ef, B, N)
        se = dict(se, x=mv(ef))
    pre = "embedding_layer.edge_embedder"
    if (opts.fused_embed_bwd and fused_embed() and G is not None and se.get("regen") is not None
            and all(f"{pre}.{l}.{t}" in G for l in (0, 2, 4) for t in ("weight", "bias"))):
        # the fused forward's saves: LayerNorm backward + both gated dX products in one launch, then the three weight (and
        # bias) gradients over the pair rows in one grouped launch on the side stream
        M = se["M"]
        dh3 = empty((M, CZ), dedge); dh2 = empty((M, CZ), dedge); dh1 = empty((M, CZ), dedge)
        img = ops.edge_embed_bwd_pack(P[f"{pre}.2.weight"], P[f"{pre}.4.weight"])
        ops.edge_embed_bwd(dedge, se["h3"], se["mean"], se["rstd"], P[f"{pre}.5.weight"], se["rowscale"], se["h2"], se["h1"], img,
                           dh3, dh2, dh1, G[f"{pre}.5.weight"], G[f"{pre}.5.bias"], M, gmask2=se.get("mh2"), gmask1=se.get("mh1"))
        x = se["x"]
        items = [(mv(dh3), mv(se["h2"]), mv(G[f"{pre}.4.weight"]), G[f"{pre}.4.bias"], CZ, CZ),
                 (mv(dh2), mv(se["h1"]), mv(G[f"{pre}.2.weight"]), G[f"{pre}.2.bias"], CZ, CZ),
                 (mv(dh1), x, mv(G[f"{pre}.0.weight"]), G[f"{pre}.0.bias"], CZ, se["K0"])]
        ops.side(lambda: ops.group_dw(items, M), (dh3, dh2, dh1, se["h2"], se["h1"], x[0]), M)
        return
    mlp3_ln_bwd(P, G, pre, se, dedge)


def pair_mask(mask, B, N):
    """edge_mask[b,i,j] = m_i * m_j as a flat [B*N*N] row scale (score_network.py:185)."""
    em = empty((B * N * N,), mask)
    # rows = (b,i) each of N columns: out[b,i,j] = mask[b,j] * mask[b,i]
    src = mask.reshape(B, 1, N).expand(B, N, N).contiguous()      # plumbing (masks only)
    lib().call("fd_rowscale", src, N, mask.reshape(-1), em, N, B * N, N)
    return em


def _adjacent_view(a, b, shape, *more):
    """One tensor over a, b (, more...) when each starts where its predecessor ends in the same storage (16-byte aligned
    start), else None."""
    ts = (a, b) + more
    for x, y in zip(ts[:-1], ts[1:]):
        if not (x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype and x.device == y.device
                and x.untyped_storage().data_ptr() == y.untyped_storage().data_ptr()
                and x.storage_offset() + x.numel() == y.storage_offset()):
            return None
    if a.data_ptr() % 16 != 0:
        return None
    strides, st = [], 1
    for d in reversed(shape):
        strides.append(st)
        st *= d
    return a.detach().as_strided(shape, tuple(reversed(strides)))


_PROJ = ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")


def _proj_views(P, pre):
    """([6816, 256] weight, [6816] bias) over IPA's four projections of s when they lie back to back, else None."""
    if not opts.proj_merge:          # (needs the back-to-back layout of optim.FlatAdam(adjacent=...))
        return None
    try:
        W = _adjacent_view(*(P[f"{pre}.{n}.weight"] for n in _PROJ[:2]), (LDP, CS), *(P[f"{pre}.{n}.weight"] for n in _PROJ[2:]))
        b = _adjacent_view(*(P[f"{pre}.{n}.bias"] for n in _PROJ[:2]), (LDP,), *(P[f"{pre}.{n}.bias"] for n in _PROJ[2:]))
    except KeyError:
        return None
    return None if W is None or b is None else (W, b)


def _joined(a, b, shape):
    v = _adjacent_view(a, b, shape)
    return v if v is not None else torch.cat([a, b], 0).contiguous()


# --------------------------------------------------------------------------- IPA
def ipa_w40(P, pre, cache=None):
    """[linear_b ; down_z] of an IPA block as one [40, 128] operand + [40] bias: a view when the two parameters lie back to
    back (optim.FlatAdam with ScoreNetwork.flat_layout_groups()), a tiny pack otherwise (once per trajectory with a cache)."""
    if cache is not None and ("W40", pre) in cache:
        return cache[("W40", pre)]
    W40 = _joined(P[f"{pre}.linear_b.weight"], P[f"{pre}.down_z.weight"], (ZB, CZ))
    b40 = _joined(P[f"{pre}.linear_b.bias"], P[f"{pre}.down_z.bias"], (ZB,))
    if cache is not None:
        cache[("W40", pre)] = (W40, b40)
    return W40, b40


def ipa_fwd(P, pre, s, z, quat, trans, mask, B, N, cache=None, zb=None, out_view=None, save=True):
    """x1 = s + mask * IPA(s, z, T).  s: matrix view [R,256].  zb: [P,40] = W40 z + b40 when the kernel that produced z
    already formed it (the previous block's fused edge transition)."""
    dev = z
    R, Pn = B * N, B * N * N
    proj = empty((R, LDP), dev)
    names = ("linear_q", "linear_kv", "linear_q_points", "linear_kv_points")     # column blocks 0, 2048, 6144, 6336
    if cache is not None:
        # static weights (sampling): the four projections of s are one GEMM over the row-concatenated weight
        if ("Wproj", pre) not in cache:
            cache[("Wproj", pre)] = (torch.cat([P[f"{pre}.{n}.weight"] for n in names], 0).contiguous(),
                                     torch.cat([P[f"{pre}.{n}.bias"] for n in names], 0).contiguous())
        Wp, bp = cache[("Wproj", pre)]
        ops.linear(s, mv(Wp), bp, mv(proj), R, LDP, CS)
    elif _proj_views(P, pre) is not None:
        # the four projection weights lie back to back (optim.FlatAdam + flat_layout_groups): one GEMM as in sampling
        Wp, bp = _proj_views(P, pre)
        ops.linear(s, mv(Wp), bp, mv(proj), R, LDP, CS)
    else:
        ops.linear(s, mv(P[f"{pre}.linear_q.weight"]), P[f"{pre}.linear_q.bias"], (proj, 0, LDP), R, H * C, CS)
        ops.linear(s, mv(P[f"{pre}.linear_kv.weight"]), P[f"{pre}.linear_kv.bias"], (proj, 2048, LDP), R, 2 * H * C, CS)
        ops.linear(s, mv(P[f"{pre}.linear_q_points.weight"]), P[f"{pre}.linear_q_points.bias"], (proj, 6144, LDP), R, 192, CS)
        ops.linear(s, mv(P[f"{pre}.linear_kv_points.weight"]), P[f"{pre}.linear_kv_points.bias"], (proj, 6336, LDP), R, 480, CS)
    qp = empty((R, H, PQ * 3), dev); kp = empty((R, H, PQ * 3), dev); vp = empty((R, H, PV * 3), dev)
    train = bool(save)             # a backward pass will read the probabilities A (and the [B, 8, 24, N] key-point copy)
    flash = (opts.flash_ipa and B * ((N + 15) // 16) >= opts.flash_ipa_min_tiles and N <= 1024
             and (lib().is_device or opts.flash_ipa_min_tiles <= 0))
    # a long lone backbone (inference): too few query tiles for the kernel above, but enough keys to split them over 4 blocks
    split = int(opts.flash_ipa_splits) if (opts.flash_ipa and not flash and not train and opts.flash_ipa_split_min_n <= N <= 1024
                  and (lib().is_device or opts.flash_ipa_split_min_n <= 16)) else 1      # (<= 16: the interpreter tests)
    flash = flash or split > 1
    # which backward will run is decided HERE, with the forward's options and tile count, and recorded for ipa_bwd: the saved
    # tensors (the key-point copy below) are the ones that path reads, whatever the options say by then
    flash_bwd = bool(train and opts.flash_ipa_bwd and B * ((N + 15) // 16) >= opts.flash_ipa_bwd_min_tiles and N <= 1024
                     and (lib().is_device or opts.flash_ipa_bwd_min_tiles <= 0))
    # (the copy serves fd_ipa_attn_fwd/bwd only: the unfused softmax kernels and both flash kernels read kp)
    kpT = empty((B, H, PQ * 3, N), dev) if opts.fused_ipa_attn and ((not flash) or (train and not flash_bwd)) else None
    par = (not train) and not flash            # sampling, launch sequence: independent launches on a second graph branch
    ops.fork(lambda: lib().call("fd_ipa_points_fwd", proj, quat, trans, qp, kp, vp, kpT, N, R, H, C, PQ, PV), proj, par)
    W40, b40 = ipa_w40(P, pre, cache)
    L = lib()
    feats = empty((R, LDF), dev)
    if zb is None:
        zb = empty((Pn, ZB), dev)
        ops.linear(mv(z), mv(W40), b40, mv(zb), Pn, ZB, CZ)
    fused_attn = opts.fused_ipa_attn
    if flash:
        # one launch: q k^T, logits, softmax, a v, a v_pts, o_pt (+ norm), o_pair; the probabilities are written out only for
        # the backward pass (training)
        A = empty((B, H, N, N), dev) if train else None
        if split > 1:
            part = empty((split * R * H * 328,), dev)
            L.call("fd_ipa_flash_fwd_split", proj, zb, qp, kp, vp, P[f"{pre}.head_weights"], mask, quat, trans, feats, None, B, N,
                   opts.flash_ipa_hpb or 4, split, part)
        else:
            L.call("fd_ipa_flash_fwd", proj, zb, qp, kp, vp, P[f"{pre}.head_weights"], mask, quat, trans, feats, A, B, N,
                   opts.flash_ipa_hpb)
    else:
        A = empty((B, H, N, N), dev)
        L.gemm(proj, proj, A, N, N, C, (LDP, 1), (1, LDP), N, b_off=2048, batch=B * H, bdiv=H,
               a_bs=(N * LDP, C), b_bs=(N * LDP, 2 * C), c_bs=(H * N * N, N * N), alpha=math.sqrt(1.0 / (3 * C)))
        if par:
            ops.join(proj)                     # the points (and whatever else the branch carried) are there
        if fused_attn:
            # logits + softmax (A in place) + o_pair (the pair part of feats) of every query row in one launch
            L.call("fd_ipa_attn_fwd", A, zb, qp, kp, kpT, P[f"{pre}.head_weights"], mask, feats, B, N)
        else:
            L.call("fd_ipa_softmax_fwd", A, zb, qp, kp, P[f"{pre}.head_weights"], mask, B, N)
        optg = empty((R, H, PV * 3), dev)

        def _pts():                            # a v_pts and o_pt: other columns of feats than a v
            L.gemm(A, vp, optg, N, PV * 3, N, (N, 1), (H * PV * 3, 1), H * PV * 3, batch=B * H, bdiv=H,
                   a_bs=(H * N * N, N * N), b_bs=(N * H * PV * 3, PV * 3), c_bs=(N * H * PV * 3, PV * 3))
            L.call("fd_ipa_opt_fwd", optg, quat, trans, feats, R)
        ops.fork(_pts, proj, par)
        L.gemm(A, proj, feats, N, C, N, (N, 1), (LDP, 1), LDF, b_off=2048 + C, batch=B * H, bdiv=H,
               a_bs=(H * N * N, N * N), b_bs=(N * LDP, 2 * C), c_bs=(N * LDF, C))
        if not fused_attn:
            L.call("fd_ipa_opair_fwd", A, zb, feats, B, N)
        if par:
            ops.join(proj)
    # out_view (a matrix view, sampling): x1 goes straight into the [R, 320] buffer whose columns 256.. take skip_embed -- the
    # operand of the first transformer layer's in_proj, which then normalises the first 256 columns itself (trunk.forward)
    x1 = empty((R, CS), dev) if out_view is None else out_view[0]
    ops.linear(mv(feats), mv(P[f"{pre}.linear_out.weight"]), P[f"{pre}.linear_out.bias"], mv(x1) if out_view is None else out_view,
               R, CS, LDF, rowscale=mask, resid=s)
    ops.STATS["ipa_flash_fwd" if flash else "ipa_sequence_fwd"] += 1
    sv = dict(s=s, z=z, quat=quat, trans=trans, mask=mask, proj=proj, qp=qp, kp=kp, kpT=kpT, vp=vp, W40=W40, b40=b40, zb=zb, A=A,
              feats=feats, B=B, N=N, fused_attn=fused_attn, flash_bwd=flash_bwd)
    return x1, sv


def ipa_bwd(P, G, pre, sv, dx1, ds, dz, dframe, dz_accumulate=True, defer_dz=False):
    """dx1 [R,256] -> accumulates ds (view, +=), dz [P,128] (+= ; = when dz_accumulate is False), dframe [R,12] (+=);
    param grads into G.  defer_dz: leave dz alone and return (dzb, W40) -- the consumer of dz adds dzb W40 itself (the fused
    backward of the edge transition in front of this block); returns None otherwise."""
    B, N = sv["B"], sv["N"]
    R, Pn = B * N, B * N * N
    dev = dx1
    L = lib()
    mask = sv["mask"]
    s, z, proj, A, feats, zb = sv["s"], sv["z"], sv["proj"], sv["A"], sv["feats"], sv["zb"]
    quat, qp, kp, vp = sv["quat"], sv["qp"], sv["kp"], sv["vp"]
    # residual branch: ds += dx1
    dm = empty((R, CS), dev)                     # d(ipa linear_out output) = mask * dx1
    L.call("fd_rowscale", dx1, CS, mask, dm, CS, R, CS)
    ops.add_view(ds, mv(dx1), R, CS)
    _lin_grads(G, f"{pre}.linear_out.weight", f"{pre}.linear_out.bias", mv(dm), mv(feats), R, CS, LDF)
    dfeats = empty((R, LDF), dev)
    ops.linear_dx(mv(dm), mv(P[f"{pre}.linear_out.weight"]), mv(dfeats), R, CS, LDF)
    dproj = empty((R, LDP), dev)            # (every column is assigned: dQ | dK, dV per head | the raw point gradients)
    dqp = empty((R, H, PQ * 3), dev); dkp = empty((R, H, PQ * 3), dev)
    dhw = G[f"{pre}.head_weights"] if G is not None else zeros((H,), dev)
    hw_part = empty((R, H), dev)
    dzb = empty((Pn, ZB), dev)
    doptg = empty((R, H, PV * 3), dev)
    flash = sv["flash_bwd"]          # (decided by ipa_fwd)
    ops.STATS["ipa_flash_bwd" if flash else "ipa_sequence_bwd"] += 1
    if flash:
        # query side in one launch: dL = A (dP - D) with dP = dO V^T + dOpt vpts^T + dout . zd formed tile by tile on the MFMA
        # (no dA in HBM), dzb, dqp, head-weight gradient; dkp from dL as before
        ptdot = empty((R, H), dev)
        L.call("fd_ipa_opt_bwd_dot", dfeats, feats, quat, sv["trans"], doptg, dframe, ptdot, R)
        dA = empty((B, H, N, N), dev)          # receives dL
        keys = bool(opts.flash_ipa_keys and N <= opts.flash_ipa_keys_max_n)       # the key side in one launch too, below
        L.call("fd_ipa_flash_bwd", proj, A, zb, dfeats, feats, doptg, ptdot, qp, kp, vp, P[f"{pre}.head_weights"], sv["trans"],
               dA, dzb, dqp, None if keys else dkp, dhw, hw_part, B, N)
    else:
        # dA = dO V^T
        dA = empty((B, H, N, N), dev)
        L.gemm(dfeats, proj, dA, N, N, C, (LDF, 1), (1, LDP), N, b_off=2048 + C, batch=B * H, bdiv=H,
               a_bs=(N * LDF, C), b_bs=(N * LDP, 2 * C), c_bs=(H * N * N, N * N))
        # o_pt
        L.call("fd_ipa_opt_bwd", dfeats, feats, quat, doptg, dframe, R)
        L.gemm(doptg, vp, dA, N, N, PV * 3, (H * PV * 3, 1), (1, H * PV * 3), N, batch=B * H, bdiv=H,
               a_bs=(N * H * PV * 3, PV * 3), b_bs=(N * H * PV * 3, PV * 3), c_bs=(H * N * N, N * N), beta=True)
        # o_pair backward + softmax backward (dA becomes dLogits) + point / bias / head-weight grads
        if sv["fused_attn"]:
            L.call("fd_ipa_attn_bwd", A, dA, zb, dfeats, qp, kp, sv["kpT"], P[f"{pre}.head_weights"], dzb, dqp, dkp, dhw,
                   hw_part, B, N)
        else:
            L.call("fd_ipa_opair_bwd", A, zb, dfeats, dA, dzb, B, N)
            L.call("fd_ipa_softmax_bwd", A, dA, qp, kp, P[f"{pre}.head_weights"], dzb, dqp, dkp, dhw, hw_part, B, N)
    dvp = empty((R, H, PV * 3), dev)
    sc = math.sqrt(1.0 / (3 * C))
    keys = flash and bool(opts.flash_ipa_keys and N <= opts.flash_ipa_keys_max_n)
    ops.STATS["ipa_flash_bwd_keys" if keys else "ipa_keys_gemms"] += 1
    if keys:
        # dV = A^T dO, dvp = A^T dOpt, dK = sc dL^T Q and dkp = gamma sum_i dL (qp_i - kp_j) of every (key tile, head) in ONE launch:
        # A and dL are read once instead of three times each (fd_ipa_flash_bwd_keys)
        L.call("fd_ipa_flash_bwd_keys", A, dA, proj, dfeats, doptg, qp, kp, P[f"{pre}.head_weights"], dproj, dvp, dkp, B, N, 0)
    else:
        # dV = A^T dO ; dvp = A^T dOpt
        L.gemm(A, dfeats, dproj, N, C, N, (1, N), (LDF, 1), LDP, c_off=2048 + C, batch=B * H, bdiv=H,
               a_bs=(H * N * N, N * N), b_bs=(N * LDF, C), c_bs=(N * LDP, 2 * C))
        L.gemm(A, doptg, dvp, N, PV * 3, N, (1, N), (H * PV * 3, 1), H * PV * 3, batch=B * H, bdiv=H,
               a_bs=(H * N * N, N * N), b_bs=(N * H * PV * 3, PV * 3), c_bs=(N * H * PV * 3, PV * 3))
    # dQ = sc * dL K ; dK = sc * dL^T Q
    L.gemm(dA, proj, dproj, N, C, N, (N, 1), (LDP, 1), LDP, b_off=2048, batch=B * H, bdiv=H,
           a_bs=(H * N * N, N * N), b_bs=(N * LDP, 2 * C), c_bs=(N * LDP, C), alpha=sc)
    if not keys:
        L.gemm(dA, proj, dproj, N, C, N, (1, N), (LDP, 1), LDP, c_off=2048, batch=B * H, bdiv=H,
               a_bs=(H * N * N, N * N), b_bs=(N * LDP, C), c_bs=(N * LDP, 2 * C), alpha=sc)
    L.call("fd_ipa_points_bwd", proj, quat, dqp, dkp, dvp, dproj, dframe, R, H, C, PQ, PV)
    # z path: dz += dzb W40 (streaming kernel, W40 resident in registers) ; dW40 += dzb^T z
    if defer_dz:
        pass
    elif sv["W40"].is_contiguous():
        L.call("fd_ipa_dz_acc", dzb, sv["W40"], dz, Pn, int(bool(dz_accumulate)))
    else:
        ops.linear_dx(mv(dzb), mv(sv["W40"]), mv(dz), Pn, ZB, CZ, beta=bool(dz_accumulate))
    if G is not None:
        gW = _adjacent_view(G[f"{pre}.linear_b.weight"], G[f"{pre}.down_z.weight"], (ZB, CZ))
        gb = _adjacent_view(G[f"{pre}.linear_b.bias"], G[f"{pre}.down_z.bias"], (ZB,))
        if gW is not None and gb is not None:
            # the two gradients lie back to back in the flat gradient buffer: accumulate into them as one [40, 128] matrix
            # (on the gradient side stream: 135 us per block that nothing on the dX chain waits for)
            def _grads_zb():
                ops.linear_dw(mv(dzb), mv(z), mv(gW), Pn, ZB, CZ)
                ops.bias_grad(mv(dzb), gb, Pn, ZB)
            ops.side(_grads_zb, (dzb, z), Pn)
        else:
            dW40 = zeros((ZB, CZ), dev); db40 = zeros((ZB,), dev)
            ops.linear_dw(mv(dzb), mv(z), mv(dW40), Pn, ZB, CZ)
            ops.bias_grad(mv(dzb), db40, Pn, ZB)
            G[f"{pre}.linear_b.weight"] += dW40[:H]; G[f"{pre}.down_z.weight"] += dW40[H:]
            G[f"{pre}.linear_b.bias"] += db40[:H]; G[f"{pre}.down_z.bias"] += db40[H:]
    # projections: ds += dproj_slice W ; dW += dproj_slice^T s
    pv = _proj_views(P, pre)
    gv = _proj_views(G, pre) if G is not None else None
    if pv is not None and gv is not None:
        # one [6816, 256] weight, one gradient: a single dX GEMM over the 6816 columns and a single dW GEMM
        ops.linear_dx(mv(dproj), mv(pv[0]), ds, R, LDP, CS, beta=True)
        if not ops.queue_dw(mv(dproj), s, mv(gv[0]), R, LDP, CS, db=gv[1]):
            ops.side(lambda: ops.linear_dw(mv(dproj), s, mv(gv[0]), R, LDP, CS, db=gv[1]), (dproj, s[0]), R)
        return (dzb, sv["W40"]) if defer_dz else None
    for name, off, n in (("linear_q", 0, 2048), ("linear_kv", 2048, 4096), ("linear_q_points", 6144, 192),
                         ("linear_kv_points", 6336, 480)):
        ops.linear_dx((dproj, off, LDP), mv(P[f"{pre}.{name}.weight"]), ds, R, n, CS, beta=True)
        _lin_grads(G, f"{pre}.{name}.weight", f"{pre}.{name}.bias", (dproj, off, LDP), s, R, n, CS)
    return (dzb, sv["W40"]) if defer_dz else None


# --------------------------------------------------------------------------- LN + skip concat
def ln_skip_fwd(P, b, x1, init_node, R):
    pre = "score_model.trunk"
    dev = x1
    u0 = empty((R, TD), dev)
    mean = empty((R,), dev); rstd = empty((R,), dev)
    ops.layernorm(mv(x1), P[f"{pre}.ipa_ln_{b}.weight"], P[f"{pre}.ipa_ln_{b}.bias"], (u0, 0, TD), R, CS, save=(mean, rstd))
    ops.linear(mv(init_node), mv(P[f"{pre}.skip_embed_{b}.weight"]), P[f"{pre}.skip_embed_{b}.bias"], (u0, CS, TD), R, 64, CS)
    return u0, dict(x1=x1, mean=mean, rstd=rstd, init_node=init_node, R=R)


def ln_skip_bwd(P, G, b, sv, du0, dx1, dinit):
    """du0 [R,320] -> dx1 [R,256] (=), dinit [R,256] (+=)."""
    pre = "score_model.trunk"
    R = sv["R"]
    ops.layernorm_bwd((du0, 0, TD), mv(sv["x1"]), P[f"{pre}.ipa_ln_{b}.weight"], sv["mean"], sv["rstd"], mv(dx1), R, CS,
                      dgamma=G[f"{pre}.ipa_ln_{b}.weight"], dbeta=G[f"{pre}.ipa_ln_{b}.bias"])
    ops.linear_dx((du0, CS, TD), mv(P[f"{pre}.skip_embed_{b}.weight"]), mv(dinit), R, 64, CS, beta=True)
    _lin_grads(G, f"{pre}.skip_embed_{b}.weight", f"{pre}.skip_embed_{b}.bias", (du0, CS, TD), mv(sv["init_node"]), R, 64, CS)


# --------------------------------------------------------------------------- transformer layer
def tfmr_layer_fwd(P, pre, x, key_add, B, N, save=True, out_rowscale=None, x_ln=None, defer_out_ln=False):
    """One post-norm TransformerEncoderLayer.  Inference at sampling sizes (save False, ops.ln_linear_ok): its LayerNorms run
    inside the GEMM launches that consume them (fd_ln_gemm) -- x_ln = (t, gamma, beta, rowscale[, ln_cols, x_out]): the layer input
    is LayerNorm(t[:, :ln_cols]) | t[:, ln_cols:] of the stage in front, not yet formed (x is ignored; x_out receives it); defer_out_ln: return the pre-norm2 tensor and
    (gamma, beta, rowscale) for the consumer instead of running norm2.  Returns (y2 | t2, saves, pending-LN | None)."""
    dev = key_add
    R = B * N
    L = lib()
    qkv = empty((R, 3 * TD), dev)
    if x_ln is not None:
        # in_proj on LayerNorm(t) of the layer in front; the normalised rows are written out too (this layer's residual)
        t_in, g_in, b_in, rs_in, lnc, x = x_ln + (0, None)[len(x_ln) - 4:]
        x = empty((R, TD), dev) if x is None else x
        ops.ln_linear(t_in if isinstance(t_in, tuple) else mv(t_in), g_in, b_in, mv(P[f"{pre}.self_attn.in_proj_weight"]), P[f"{pre}.self_attn.in_proj_bias"], mv(qkv),
                      R, 3 * TD, TD, ln_rowscale=rs_in, ln_out=mv(x), ln_cols=lnc)
    else:
        ops.linear(mv(x), mv(P[f"{pre}.self_attn.in_proj_weight"]), P[f"{pre}.self_attn.in_proj_bias"], mv(qkv), R, 3 * TD, TD)
    o = empty((R, TD), dev)
    if opts.fused_seq_attn and R >= opts.seq_attn_min_rows:
        # scores + key mask + softmax + value product of every (batch, head) in one launch; the probabilities reach HBM
        # only when a backward pass will need them.  (Below ~1,000 residue rows its B x 4 x N/32 blocks are too few: a
        # lone N = 256 backbone samples 2-5 % slower with it; B = 32 x N = 128 is 2 % faster, the training step 1 %.)
        A = empty((B, TH, N, N), dev) if save else None
        L.call("fd_seq_attn_fwd", qkv, key_add, o, A, 1.0 / math.sqrt(THD), B, N)
    else:
        A = empty((B, TH, N, N), dev)
        L.gemm(qkv, qkv, A, N, N, THD, (3 * TD, 1), (1, 3 * TD), N, b_off=TD, batch=B * TH, bdiv=TH,
               a_bs=(N * 3 * TD, THD), b_bs=(N * 3 * TD, THD), c_bs=(TH * N * N, N * N), alpha=1.0 / math.sqrt(THD))
        L.call("fd_row_softmax_fwd", A, key_add, B * TH * N, N, TH * N)
        L.gemm(A, qkv, o, N, THD, N, (N, 1), (3 * TD, 1), TD, b_off=2 * TD, batch=B * TH, bdiv=TH,
               a_bs=(TH * N * N, N * N), b_bs=(N * 3 * TD, THD), c_bs=(N * TD, THD))
    t1 = empty((R, TD), dev)
    ops.linear(mv(o), mv(P[f"{pre}.self_attn.out_proj.weight"]), P[f"{pre}.self_attn.out_proj.bias"], mv(t1), R, TD, TD, resid=mv(x))
    fold = (not save) and ops.ln_linear_ok(mv(t1), mv(P[f"{pre}.linear1.weight"]), R, TD, TD)
    y1 = empty((R, TD), dev); m1 = r1 = None
    f = empty((R, TD), dev)
    if fold:
        # norm1 inside linear1's launch; y1 (the residual of linear2) is written by the same launch
        ops.ln_linear(mv(t1), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"], mv(P[f"{pre}.linear1.weight"]),
                      P[f"{pre}.linear1.bias"], mv(f), R, TD, TD, relu=True, ln_out=mv(y1))
    else:
        m1 = empty((R,), dev); r1 = empty((R,), dev)
        ops.layernorm(mv(t1), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"], mv(y1), R, TD, save=(m1, r1))
        ops.linear(mv(y1), mv(P[f"{pre}.linear1.weight"]), P[f"{pre}.linear1.bias"], mv(f), R, TD, TD, relu=True)
    t2 = empty((R, TD), dev)
    ops.linear(mv(f), mv(P[f"{pre}.linear2.weight"]), P[f"{pre}.linear2.bias"], mv(t2), R, TD, TD, resid=mv(y1))
    if fold and defer_out_ln:
        return t2, None, (t2, P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"], out_rowscale)
    y2 = empty((R, TD), dev); m2 = empty((R,), dev); r2 = empty((R,), dev)
    ops.layernorm(mv(t2), P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"], mv(y2), R, TD, rowscale=out_rowscale, save=(m2, r2))
    return y2, dict(x=x, qkv=qkv, A=A, o=o, t1=t1, m1=m1, r1=r1, y1=y1, f=f, t2=t2, m2=m2, r2=r2, B=B, N=N), None


def tfmr_layer_bwd(P, G, pre, sv, dy2):
    """returns dx [R,320]."""
    B, N = sv["B"], sv["N"]
    R = B * N
    dev = dy2
    L = lib()
    dt2 = empty((R, TD), dev)
    ops.layernorm_bwd(mv(dy2), mv(sv["t2"]), P[f"{pre}.norm2.weight"], sv["m2"], sv["r2"], mv(dt2), R, TD,
                      dgamma=G[f"{pre}.norm2.weight"], dbeta=G[f"{pre}.norm2.bias"])
    _lin_grads(G, f"{pre}.linear2.weight", f"{pre}.linear2.bias", mv(dt2), mv(sv["f"]), R, TD, TD)
    df = empty((R, TD), dev)
    ops.linear_dx(mv(dt2), mv(P[f"{pre}.linear2.weight"]), mv(df), R, TD, TD, gate=mv(sv["f"]))
    _lin_grads(G, f"{pre}.linear1.weight", f"{pre}.linear1.bias", mv(df), mv(sv["y1"]), R, TD, TD)
    dy1 = empty((R, TD), dev)  # dy1 = dt2 (residual) + df W1 (operands of side-stream gradients stay read-only)
    ops.linear_dx(mv(df), mv(P[f"{pre}.linear1.weight"]), mv(dy1), R, TD, TD, resid=mv(dt2))
    dt1 = empty((R, TD), dev)
    ops.layernorm_bwd(mv(dy1), mv(sv["t1"]), P[f"{pre}.norm1.weight"], sv["m1"], sv["r1"], mv(dt1), R, TD,
                      dgamma=G[f"{pre}.norm1.weight"], dbeta=G[f"{pre}.norm1.bias"])
    _lin_grads(G, f"{pre}.self_attn.out_proj.weight", f"{pre}.self_attn.out_proj.bias", mv(dt1), mv(sv["o"]), R, TD, TD)
    do = empty((R, TD), dev)
    ops.linear_dx(mv(dt1), mv(P[f"{pre}.self_attn.out_proj.weight"]), mv(do), R, TD, TD)
    qkv, A = sv["qkv"], sv["A"]
    dqkv = empty((R, 3 * TD), dev)
    sc = 1.0 / math.sqrt(THD)
    if opts.fused_seq_attn_bwd and not L.exact_f32 and (N <= 256 or (B * N >= 4096 and N <= 1024)):
        # (a long lone backbone -- 64 blocks walking 32 tiles each -- keeps the batched GEMMs: 62 against 33 us at B=1 x N=512)
        # dQ, dK, dV of every (batch, head) in ONE launch from the saved probabilities and the saved output (fd_seq_attn_bwd):
        # dA / dS never reach HBM (five launches before: two batched GEMMs, the row-softmax backward, two batched GEMMs)
        L.call("fd_seq_attn_bwd", qkv, A, do, sv["o"], dqkv, sc, B, N)
    else:
        dA = empty((B, TH, N, N), dev)
        # dA = do V^T ; dV = A^T do
        L.gemm(do, qkv, dA, N, N, THD, (TD, 1), (1, 3 * TD), N, b_off=2 * TD, batch=B * TH, bdiv=TH,
               a_bs=(N * TD, THD), b_bs=(N * 3 * TD, THD), c_bs=(TH * N * N, N * N))
        L.gemm(A, do, dqkv, N, THD, N, (1, N), (TD, 1), 3 * TD, c_off=2 * TD, batch=B * TH, bdiv=TH,
               a_bs=(TH * N * N, N * N), b_bs=(N * TD, THD), c_bs=(N * 3 * TD, THD))
        L.call("fd_row_softmax_bwd", A, dA, B * TH * N, N)
        L.gemm(dA, qkv, dqkv, N, THD, N, (N, 1), (3 * TD, 1), 3 * TD, b_off=TD, batch=B * TH, bdiv=TH,
               a_bs=(TH * N * N, N * N), b_bs=(N * 3 * TD, THD), c_bs=(N * 3 * TD, THD), alpha=sc)
        L.gemm(dA, qkv, dqkv, N, THD, N, (1, N), (3 * TD, 1), 3 * TD, c_off=TD, batch=B * TH, bdiv=TH,
               a_bs=(TH * N * N, N * N), b_bs=(N * 3 * TD, THD), c_bs=(N * 3 * TD, THD), alpha=sc)
    _lin_grads(G, f"{pre}.self_attn.in_proj_weight", f"{pre}.self_attn.in_proj_bias", mv(dqkv), mv(sv["x"]), R, 3 * TD, TD)
    dx = empty((R, TD), dev)  # dx = dt1 (residual) + dqkv W_in
    ops.linear_dx(mv(dqkv), mv(P[f"{pre}.self_attn.in_proj_weight"]), mv(dx), R, 3 * TD, TD, resid=mv(dt1))
    return dx


# --------------------------------------------------------------------------- post-tfmr + node transition
def post_node_fwd(P, b, u2, u0, mask, R, u2_ln=None, defer_ln=False):
    """n2 = u0[:, :256] + W_post u2 ; n3 = mask * LN(n2 + W3 relu(W2 relu(W1 n2))).  u2_ln = (t, gamma, beta, rowscale): u2 is
    LayerNorm(t) of the last transformer layer, formed inside post_tfmr's launch (tfmr_layer_fwd defer_out_ln)."""
    pre = "score_model.trunk"
    dev = u0
    n2 = empty((R, CS), dev)
    if u2_ln is not None:
        t_in, g_in, b_in, rs_in = u2_ln
        ops.ln_linear(mv(t_in), g_in, b_in, mv(P[f"{pre}.post_tfmr_{b}.weight"]), P[f"{pre}.post_tfmr_{b}.bias"], mv(n2), R, CS, TD,
                      resid=(u0, 0, TD), ln_rowscale=rs_in)
    else:
        ops.linear(mv(u2), mv(P[f"{pre}.post_tfmr_{b}.weight"]), P[f"{pre}.post_tfmr_{b}.bias"], mv(n2), R, CS, TD, resid=(u0, 0, TD))
    nt = f"{pre}.node_transition_{b}"
    h1 = empty((R, CS), dev); h2 = empty((R, CS), dev); t = empty((R, CS), dev); n3 = empty((R, CS), dev)
    mean = empty((R,), dev); rstd = empty((R,), dev)
    ops.linear(mv(n2), mv(P[f"{nt}.linear_1.weight"]), P[f"{nt}.linear_1.bias"], mv(h1), R, CS, CS, relu=True)
    ops.linear(mv(h1), mv(P[f"{nt}.linear_2.weight"]), P[f"{nt}.linear_2.bias"], mv(h2), R, CS, CS, relu=True)
    ops.linear(mv(h2), mv(P[f"{nt}.linear_3.weight"]), P[f"{nt}.linear_3.bias"], mv(t), R, CS, CS, resid=mv(n2))
    if defer_ln:
        # sampling: the transition's LayerNorm (x node mask) runs inside the launch of its first GEMM consumer, which writes n3
        return None, (t, P[f"{nt}.ln.weight"], P[f"{nt}.ln.bias"], mask, 0, n3)
    ops.layernorm(mv(t), P[f"{nt}.ln.weight"], P[f"{nt}.ln.bias"], mv(n3), R, CS, rowscale=mask, save=(mean, rstd))
    return n3, dict(u2=u2, n2=n2, h1=h1, h2=h2, t=t, mean=mean, rstd=rstd, mask=mask, R=R)


def post_node_bwd(P, G, b, sv, dn3, du0):
    """dn3 [R,256] -> returns du2 [R,320]; du0[:, :256] += dn2."""
    pre = "score_model.trunk"
    nt = f"{pre}.node_transition_{b}"
    R = sv["R"]
    dev = dn3
    dt = empty((R, CS), dev)
    ops.layernorm_bwd(mv(dn3), mv(sv["t"]), P[f"{nt}.ln.weight"], sv["mean"], sv["rstd"], mv(dt), R, CS,
                      rowscale=sv["mask"], dgamma=G[f"{nt}.ln.weight"], dbeta=G[f"{nt}.ln.bias"])
    _lin_grads(G, f"{nt}.linear_3.weight", f"{nt}.linear_3.bias", mv(dt), mv(sv["h2"]), R, CS, CS)
    dh2 = empty((R, CS), dev)
    ops.linear_dx(mv(dt), mv(P[f"{nt}.linear_3.weight"]), mv(dh2), R, CS, CS, gate=mv(sv["h2"]))
    _lin_grads(G, f"{nt}.linear_2.weight", f"{nt}.linear_2.bias", mv(dh2), mv(sv["h1"]), R, CS, CS)
    dh1 = empty((R, CS), dev)
    ops.linear_dx(mv(dh2), mv(P[f"{nt}.linear_2.weight"]), mv(dh1), R, CS, CS, gate=mv(sv["h1"]))
    _lin_grads(G, f"{nt}.linear_1.weight", f"{nt}.linear_1.bias", mv(dh1), mv(sv["n2"]), R, CS, CS)
    dn2 = empty((R, CS), dev)  # dn2 = dt (residual) + dh1 W1
    ops.linear_dx(mv(dh1), mv(P[f"{nt}.linear_1.weight"]), mv(dn2), R, CS, CS, resid=mv(dt))
    _lin_grads(G, f"{pre}.post_tfmr_{b}.weight", f"{pre}.post_tfmr_{b}.bias", mv(dn2), mv(sv["u2"]), R, CS, TD)
    du2 = empty((R, TD), dev)
    ops.linear_dx(mv(dn2), mv(P[f"{pre}.post_tfmr_{b}.weight"]), mv(du2), R, CS, TD)
    ops.add_view((du0, 0, TD), mv(dn2), R, CS)
    return du2
