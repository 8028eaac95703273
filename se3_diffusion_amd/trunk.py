"""Remaining ScoreNetwork stages (backbone update, edge transition, heads) and the
whole-network forward / backward drivers.  See network.py for conventions."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import hip, ops
from .trace import rng
from . import network as nw
from .network import CS, CZ, TD, H, _lin_grads
from .ops import empty, zeros, mv, lib
from .options import opts

EH = 384  # edge-transition hidden = c_z + 2 * (c_s // 2)
CE = 128  # edge-transition node embedding (c_s // 2)


# --------------------------------------------------------------------------- backbone update
def bb_update_fwd(P, b, n3, dmask, quat, trans, R):
    pre = f"score_model.trunk.bb_update_{b}.linear"
    dev = n3
    upd = empty((R, 6), dev); q2 = empty((R, 4), dev); t2 = empty((R, 3), dev)
    lib().call("fd_bb_update_fwd", n3, CS, CS, dmask, P[f"{pre}.weight"], P[f"{pre}.bias"], quat, trans, upd, q2, t2, R)
    return q2, t2, dict(n3=n3, dmask=dmask, quat=quat, upd=upd, R=R)


def bb_update_bwd(P, G, b, sv, dq2, dt2, dframe, dn3):
    """(dq2, dt2) grads of the updated frame; dframe [R,12] = grads of the INPUT frame collected by the IPA
    kernels.  Returns (dq, dt) of the input frame; accumulates dn3 (+=)."""
    pre = f"score_model.trunk.bb_update_{b}.linear"
    R = sv["R"]
    dev = dq2
    dq = empty((R, 4), dev); dt = empty((R, 3), dev); dupd = empty((R, 6), dev); dupd_s = empty((R, 6), dev)
    lib().call("fd_bb_update_bwd", dq2, dt2, dframe, sv["dmask"], sv["upd"], sv["quat"], dq, dt, dupd, dupd_s, R)
    # upd = W6 (n3 * d) + b6
    if G is not None:
        def _grads():
            ops.linear_dw(mv(dupd_s), mv(sv["n3"]), mv(G[f"{pre}.weight"]), R, 6, CS)
            ops.bias_grad(mv(dupd), G[f"{pre}.bias"], R, 6)
        ops.side(_grads, (dupd_s, dupd, sv["n3"]), R)
    ops.linear_dx(mv(dupd_s), mv(P[f"{pre}.weight"]), mv(dn3), R, 6, CS, beta=True)
    return dq, dt


# --------------------------------------------------------------------------- edge transition
def _edge_mlp_image(P, pre, cache, backward=False, W40=None):
    """bf16-plane weight image of the fused kernel: packed once per forward in training (the weights change every
    step), once per trajectory in sampling (cache).  W40: the next IPA block's [linear_b ; down_z] (fourth layer)."""
    key = ("et_img", pre, backward, W40 is not None)
    if cache is not None and key in cache:
        return cache[key]
    img = ops.edge_mlp_pack(P[f"{pre}.trunk.0.weight"], P[f"{pre}.trunk.2.weight"], P[f"{pre}.final_layer.weight"],
                            backward=backward, W40=W40)
    if cache is not None:
        cache[key] = img
    return img


def fused_edge():
    """The fused edge-transition kernel computes in split-bf16 (fp32-accurate): off in exact-fp32 mode."""
    return opts.fused_edge and not lib().exact_f32


def et_folds_node_terms(cache, save):
    """sampling: edge_transition_fwd forms the per-residue terms with ONE GEMM of n3 (against a matrix folded once per trajectory)"""
    return cache is not None and not save and opts.fold_node_terms and fused_edge()


def edge_transition_fwd(P, b, n3, z, emask, B, N, save=True, cache=None, zb_next=None, n3_ln=None, after_terms=None):
    """z' = emask * LN(W_f (relu(W_2 relu(W_1 x)) + x) + b_f), x = [z | e_i | e_j], e = W_init n3 -- the whole pair-level
    chain in ONE launch (fd_edge_mlp): h1 / h2 never reach HBM unless the backward needs them (save).
    zb_next = (W40 [40,128], b40 [40]) of the next block's IPA: its pair projection zb = W40 z' + b40 is formed by the same
    launch from z' in registers.  Returns (z', saved, zb or None)."""
    if not fused_edge():
        return edge_transition_fwd_unfused(P, b, n3, z, emask, B, N) + (None,)
    pre = f"score_model.trunk.edge_transition_{b}"
    dev = z
    R, Pn = B * N, B * N * N
    W1, Wf = P[f"{pre}.trunk.0.weight"], P[f"{pre}.final_layer.weight"]
    kw = {}
    assert n3_ln is None or et_folds_node_terms(cache, save)
    if et_folds_node_terms(cache, save):
        # static weights (sampling): the per-residue terms P1 | Q1 | Pf | Qf = [W1_i; W1_j; Wf_i; Wf_j] (W_init n3 + b_init)
        # (+ b1, b_f) are ONE GEMM of n3 against a matrix folded once per trajectory (5 launches -> 1); the fused kernel
        # reads the four column blocks of its [R, 1024] output through their row stride
        key = ("et_fold", pre)
        if key not in cache:
            Wi, bi = P[f"{pre}.initial_embed.weight"], P[f"{pre}.initial_embed.bias"]
            Wcat = torch.cat([W1[:, CZ:CZ + CE], W1[:, CZ + CE:], Wf[:, CZ:CZ + CE], Wf[:, CZ + CE:]], 0).contiguous()
            Wfold = empty((2 * EH + 2 * CZ, CS), dev)
            ops.linear(mv(Wcat), mv(Wi.t().contiguous()), None, mv(Wfold), 2 * EH + 2 * CZ, CS, CE)
            bvec = torch.cat([torch.zeros_like(P[f"{pre}.trunk.0.bias"]), P[f"{pre}.trunk.0.bias"],
                              torch.zeros_like(P[f"{pre}.final_layer.bias"]), P[f"{pre}.final_layer.bias"]]).contiguous()
            bfold = empty((1, 2 * EH + 2 * CZ), dev)
            ops.linear(mv(bi.view(1, CE)), mv(Wcat), bvec, mv(bfold), 1, 2 * EH + 2 * CZ, CE)
            cache[key] = (Wfold, bfold.view(-1))
        Wfold, bfold = cache[key]
        LDT = 2 * EH + 2 * CZ
        PQ = empty((R, LDT), dev)
        if n3_ln is not None:
            # n3 = mask * LayerNorm(t) of the node transition, formed (and written out for the other consumers) by this launch
            t_in, g_in, b_in, rs_in, _, n3_out = n3_ln
            ops.ln_linear(mv(t_in), g_in, b_in, mv(Wfold), bfold, mv(PQ), R, LDT, CS, ln_rowscale=rs_in, ln_out=mv(n3_out))
        else:
            ops.linear(mv(n3), mv(Wfold), bfold, mv(PQ), R, LDT, CS)
        if after_terms is not None:
            after_terms()          # (n3 exists from here on: the caller forks the backbone update beside the pair-level launch)
        P1, Q1, Pf, Qf = PQ[:, 0:], PQ[:, EH:], PQ[:, 2 * EH:], PQ[:, 2 * EH + CZ:]
        kw = dict(ld_pq=LDT, ld_pqf=LDT)
        e = None
    else:
        e = empty((R, CE), dev)
        ops.linear(mv(n3), mv(P[f"{pre}.initial_embed.weight"]), P[f"{pre}.initial_embed.bias"], mv(e), R, CE, CS)
        # node halves of the two concat-linears: W [z | e_i | e_j] = W_z z + (W_i e)_i + (W_j e)_j
        P1 = empty((R, EH), dev); Q1 = empty((R, EH), dev); Pf = empty((R, CZ), dev); Qf = empty((R, CZ), dev)
        ops.linear(mv(e), (W1, CZ, EH), None, mv(P1), R, EH, CE)
        ops.linear(mv(e), (W1, CZ + CE, EH), P[f"{pre}.trunk.0.bias"], mv(Q1), R, EH, CE)
        ops.linear(mv(e), (Wf, CZ, EH), None, mv(Pf), R, CZ, CE)
        ops.linear(mv(e), (Wf, CZ + CE, EH), P[f"{pre}.final_layer.bias"], mv(Qf), R, CZ, CE)
    zb = None
    use_zb = (opts.zb_from_edge and zb_next is not None and zb_next[0].is_contiguous() and zb_next[0].data_ptr() % 16 == 0
              and zb_next[1].data_ptr() % 16 == 0)
    img = _edge_mlp_image(P, pre, cache, W40=zb_next[0] if use_zb else None)
    z2 = empty((Pn, CZ), dev)
    if save:
        # h2z = h2 + [z | 0 | 0]: the input of the final layer (ipa_pytorch.py:231), i.e. the operand of its weight gradient;
        # mh1 / mh2: sign bits of h1 / h2 (48 B per row each) -- the ReLU gates of the backward's dX kernel, which reads no h1 / h2
        h1 = empty((Pn, EH), dev); h2 = empty((Pn, EH), dev); y = empty((Pn, CZ), dev)
        mean = empty((Pn,), dev); rstd = empty((Pn,), dev)
        mh1 = empty((Pn, 12), dev, torch.int32); mh2 = empty((Pn, 12), dev, torch.int32)
        kw = dict(save1=h1, save2=h2, y=y, mean=mean, rstd=rstd, mask1=mh1, mask2=mh2)
    if use_zb:
        zb = empty((Pn, nw.ZB), dev)
        kw.update(zb_out=zb, zb_bias=zb_next[1])
    ops.edge_mlp(z, img, z2, Pn, N, p1=P1, q1=Q1, bias2=P[f"{pre}.trunk.2.bias"], pf=Pf, qf=Qf,
                 gamma=P[f"{pre}.layer_norm.weight"], beta=P[f"{pre}.layer_norm.bias"], rowscale=emask, **kw)
    if not save:
        return z2, None, zb
    return z2, dict(n3=n3, z=z, e=e, h1=h1, h2=h2, y=y, mean=mean, rstd=rstd, emask=emask, B=B, N=N,
                    mh1=mh1, mh2=mh2, h2_has_z=True), zb


def edge_transition_fwd_unfused(P, b, n3, z, emask, B, N):
    """z' = emask * LN(W_f (relu(W_2 relu(W_1 x)) + x) + b_f), x = [z | e_i | e_j], e = W_init n3.
    The concat is never materialised: W x = W[:, :128] z + (W[:,128:256] e)_i + (W[:,256:] e)_j."""
    pre = f"score_model.trunk.edge_transition_{b}"
    dev = z
    R, Pn = B * N, B * N * N
    W1, Wf = P[f"{pre}.trunk.0.weight"], P[f"{pre}.final_layer.weight"]
    e = empty((R, CE), dev)
    ops.linear(mv(n3), mv(P[f"{pre}.initial_embed.weight"]), P[f"{pre}.initial_embed.bias"], mv(e), R, CE, CS)
    P1 = empty((R, EH), dev); Q1 = empty((R, EH), dev)
    ops.linear(mv(e), (W1, CZ, EH), None, mv(P1), R, EH, CE)
    ops.linear(mv(e), (W1, CZ + CE, EH), P[f"{pre}.trunk.0.bias"], mv(Q1), R, EH, CE)
    h1 = empty((Pn, EH), dev)
    ops.linear(mv(z), (W1, 0, EH), None, mv(h1), Pn, EH, CZ, relu=True, pair=(P1, Q1, EH, N))
    h2 = empty((Pn, EH), dev)
    ops.linear(mv(h1), mv(P[f"{pre}.trunk.2.weight"]), P[f"{pre}.trunk.2.bias"], mv(h2), Pn, EH, EH, relu=True)
    Pf = empty((R, CZ), dev); Qf = empty((R, CZ), dev)
    ops.linear(mv(e), (Wf, CZ, EH), None, mv(Pf), R, CZ, CE)
    ops.linear(mv(e), (Wf, CZ + CE, EH), P[f"{pre}.final_layer.bias"], mv(Qf), R, CZ, CE)
    y = empty((Pn, CZ), dev)
    ops.linear(mv(h2), mv(Wf), None, mv(y), Pn, CZ, EH)
    ops.linear(mv(z), (Wf, 0, EH), None, mv(y), Pn, CZ, CZ, pair=(Pf, Qf, CZ, N), beta=True)
    z2 = empty((Pn, CZ), dev); mean = empty((Pn,), dev); rstd = empty((Pn,), dev)
    ops.layernorm(mv(y), P[f"{pre}.layer_norm.weight"], P[f"{pre}.layer_norm.bias"], mv(z2), Pn, CZ, rowscale=emask,
                  save=(mean, rstd))
    return z2, dict(n3=n3, z=z, e=e, h1=h1, h2=h2, y=y, mean=mean, rstd=rstd, emask=emask, B=B, N=N)


def edge_transition_bwd(P, G, b, sv, dz2, dz, dn3, dzb_next=None, flush_behind_launch=False):
    """dz2 [P,128] (gradient of the transition's output; None when dzb_next carries all of it) -> dz [P,128] (=), dn3 (+=).
    dzb_next = (dzb [P,40], W40 [40,128]) of the IPA block BEHIND this transition: its pair-projection term dzb W40 still has
    to be added to dz2 (the fused backward kernel does it in its prologue; the other paths through fd_ipa_dz_acc here)."""
    pre = f"score_model.trunk.edge_transition_{b}"
    B, N = sv["B"], sv["N"]
    R, Pn = B * N, B * N * N
    dev = sv["z"]
    L = lib()
    W1, W2, Wf = P[f"{pre}.trunk.0.weight"], P[f"{pre}.trunk.2.weight"], P[f"{pre}.final_layer.weight"]
    z, e, h1, h2 = sv["z"], sv["e"], sv["h1"], sv["h2"]
    # which backward runs is decided by what the FORWARD saved, not by the options of this moment: the fused dX kernel gates on the
    # packed masks the fused forward wrote, and that forward's h2 save carries the residual z (h2_has_z) -- only the fused backward
    # can consume it; a forward that ran unfused is followed by the unfused backward
    fused = sv.get("mh1") is not None and (fused_edge() or bool(sv.get("h2_has_z")))
    h2_has_z = bool(sv.get("h2_has_z"))      # the fused forward saved h2 + [z | 0 | 0] (the operand of dWf), not h2
    assert fused or not h2_has_z, "a fused edge-transition forward needs the fused backward (its h2 save carries z)"
    fused_ln = fused and opts.fused_ln_bwd and (dzb_next is None or dzb_next[1].is_contiguous())
    if dzb_next is not None and not fused_ln:
        # materialise the IPA term first (streaming kernel, W40 resident in registers)
        dzb, W40 = dzb_next
        if dz2 is None:
            dz2 = empty((Pn, CZ), dev)
            L.call("fd_ipa_dz_acc", dzb, W40.contiguous(), dz2, Pn, 0)
        else:
            L.call("fd_ipa_dz_acc", dzb, W40.contiguous(), dz2, Pn, 1)
    dy = empty((Pn, CZ), dev)
    gWf = G[f"{pre}.final_layer.weight"]
    gW1 = G[f"{pre}.trunk.0.weight"]
    grouped_dw = fused and opts.grouped_pair_dw
    dh2 = dh1 = None
    if fused_ln:
        # ONE launch: (dz2 + dzb W40) -> LayerNorm backward (dgamma, dbeta) -> dy -> d2 = [h2 > 0] dy Wf -> d1 = [h1 > 0] d2 W2 ->
        # dz = dy Wf_z + d1 W1_z; dy, d2, d1 are written once for the weight gradients and the pair reductions
        dh2 = empty((Pn, EH), dev); dh1 = empty((Pn, EH), dev)
        img = ops.edge_mlp_pack_bwd(Wf, W2, W1, W40=dzb_next[1] if dzb_next is not None else None)
        gk = dict(gmask1=sv["mh2"], gmask2=sv["mh1"])
        ops.edge_mlp(dz2, img, dz, Pn, N, save1=dh2, save2=dh1, backward=True, ln_y=sv["y"], ln_mean=sv["mean"],
                     ln_rstd=sv["rstd"], ln_gamma=P[f"{pre}.layer_norm.weight"], ln_rowscale=sv["emask"], dy_out=dy,
                     ln_dgamma=G[f"{pre}.layer_norm.weight"], ln_dbeta=G[f"{pre}.layer_norm.bias"],
                     dzb=dzb_next[0] if dzb_next is not None else None, **gk)
        if flush_behind_launch:
            # the grouped node-level weight gradients of the block BEHIND this transition (queued by the previous iteration of
            # trunk._backward) go to the side stream HERE, behind the full-chip kernel just issued: their launch then shares the
            # chip with this block's latency-bound node-level kernels -- flushed at the end of their own block they stood, on
            # every CU, in front of this kernel (0.2-0.25 ms per block in profiles/r05_step_gap.txt: the 4-byte memset that
            # zeroes the tile counter waited 200 us for a CU)
            ops.flush_dw()
    else:
        if flush_behind_launch:
            ops.flush_dw()
        ops.layernorm_bwd(mv(dz2), mv(sv["y"]), P[f"{pre}.layer_norm.weight"], sv["mean"], sv["rstd"], mv(dy), Pn, CZ,
                          rowscale=sv["emask"], dgamma=G[f"{pre}.layer_norm.weight"], dbeta=G[f"{pre}.layer_norm.bias"])
    # y = Wf h2 + Wf[:, :128] z + Pf_i + Qf_j (+bf inside Qf)
    if not grouped_dw:
        def _grads_y():
            ops.linear_dw(mv(dy), mv(h2), mv(gWf), Pn, CZ, EH)
            if not h2_has_z:
                ops.linear_dw(mv(dy), mv(z), (gWf, 0, EH), Pn, CZ, CZ)
        ops.side(_grads_y, (dy, h2, z), Pn)
    dPf = zeros((R, CZ), dev); dQf = zeros((R, CZ), dev)
    L.call("fd_pair_reduce_acc", dy, B, N, CZ, dPf, dQf, CZ)
    def _grads_f():
        ops.linear_dw(mv(dPf), mv(e), (gWf, CZ, EH), R, CZ, CE)
        ops.linear_dw(mv(dQf), mv(e), (gWf, CZ + CE, EH), R, CZ, CE)
        ops.bias_grad(mv(dQf), G[f"{pre}.final_layer.bias"], R, CZ)
    # (both products or neither: a half-accepted pair followed by the fallback would add the accepted product twice)
    if (ops.queue_dw_ok(mv(dPf), mv(e), (gWf, CZ, EH), R, CZ, CE)
            and ops.queue_dw_ok(mv(dQf), mv(e), (gWf, CZ + CE, EH), R, CZ, CE)):
        ops.queue_dw(mv(dPf), mv(e), (gWf, CZ, EH), R, CZ, CE)
        ops.queue_dw(mv(dQf), mv(e), (gWf, CZ + CE, EH), R, CZ, CE, db=G[f"{pre}.final_layer.bias"])
    else:
        ops.side(_grads_f, (dPf, dQf, e), R)
    de = empty((R, CE), dev)
    ops.linear_dx(mv(dPf), (Wf, CZ, EH), mv(de), R, CZ, CE)
    ops.linear_dx(mv(dQf), (Wf, CZ + CE, EH), mv(de), R, CZ, CE, beta=True)
    if fused:
        if not fused_ln:
            # the dX chain in one launch: d2 = [h2 > 0] dy Wf, d1 = [h1 > 0] d2 W2, dz = dy Wf_z + d1 W1_z (fd_edge_mlp with
            # the transposed weight image); d2 / d1 are written once, for the weight-gradient GEMMs and the pair reductions
            dh2 = empty((Pn, EH), dev); dh1 = empty((Pn, EH), dev)
            gk = dict(gmask1=sv["mh2"], gmask2=sv["mh1"])
            ops.edge_mlp(dy, _edge_mlp_image(P, pre, None, backward=True), dz, Pn, N, save1=dh2, save2=dh1, backward=True, **gk)
        if grouped_dw:
            # every pair-row weight gradient of the transition in ONE grouped launch (fd_pair_dw): dW2 = d2^T h1 as three
            # 384 x 128 tiles (+ its bias gradient), dW1[:, z part] = d1^T z, dWf = dy^T (h2 + [z | 0]) stored transposed (the
            # fused forward saved that sum: no A_add)
            gW2, gb2 = G[f"{pre}.trunk.2.weight"], G[f"{pre}.trunk.2.bias"]
            items = [dict(A=(dh2, 0, EH), B=(h1, CZ * j, EH), C=(gW2, CZ * j, EH), colsum=gb2 if j == 0 else None)
                     for j in range(3)]
            items.append(dict(A=(dh1, 0, EH), B=(z, 0, CZ), C=(gW1, 0, EH)))
            it = dict(A=(h2, 0, EH), B=(dy, 0, CZ), C=(gWf, 0, EH), trans=True)
            if not h2_has_z:
                it["A_add"] = (z, 0, CZ)
            items.append(it)
            # On the side stream the grouped kernel takes 160 of the 256 CUs (32 row ranges x 5 tiles): it then runs 1.6x longer
            # but BESIDE the ~100 latency-bound node-level / IPA launches the main stream issues next, instead of holding
            # every CU while they queue behind it (27.1 -> 26.5 ms per step; 128 / 192 / 256 blocks: 26.6 / 26.6 / 27.1)
            nblk = opts.pair_dw_blocks if ops.side_active(dh2, Pn) else 0
            ops.side(lambda: ops.pair_dw(items, Pn, blocks=nblk), (dh2, dh1, h1, h2, z, dy), Pn)
        else:
            _lin_grads(G, f"{pre}.trunk.2.weight", f"{pre}.trunk.2.bias", mv(dh2), mv(h1), Pn, EH, EH)
        del dh2
    else:
        ops.linear_dx(mv(dy), (Wf, 0, EH), mv(dz), Pn, CZ, CZ)                     # dz = dy Wf_z
        dh2 = empty((Pn, EH), dev)
        ops.linear_dx(mv(dy), mv(Wf), mv(dh2), Pn, CZ, EH, gate=mv(h2))
        _lin_grads(G, f"{pre}.trunk.2.weight", f"{pre}.trunk.2.bias", mv(dh2), mv(h1), Pn, EH, EH)
        dh1 = empty((Pn, EH), dev)
        ops.linear_dx(mv(dh2), mv(W2), mv(dh1), Pn, EH, EH, gate=mv(h1))
        del dh2
    if not grouped_dw:
        ops.side(lambda: ops.linear_dw(mv(dh1), mv(z), (gW1, 0, EH), Pn, EH, CZ), (dh1, z), Pn)
    dP1 = zeros((R, EH), dev); dQ1 = zeros((R, EH), dev)
    L.call("fd_pair_reduce_acc", dh1, B, N, EH, dP1, dQ1, EH)
    def _grads_1():
        ops.linear_dw(mv(dP1), mv(e), (gW1, CZ, EH), R, EH, CE)
        ops.linear_dw(mv(dQ1), mv(e), (gW1, CZ + CE, EH), R, EH, CE)
        ops.bias_grad(mv(dQ1), G[f"{pre}.trunk.0.bias"], R, EH)
    if (ops.queue_dw_ok(mv(dP1), mv(e), (gW1, CZ, EH), R, EH, CE)
            and ops.queue_dw_ok(mv(dQ1), mv(e), (gW1, CZ + CE, EH), R, EH, CE)):
        ops.queue_dw(mv(dP1), mv(e), (gW1, CZ, EH), R, EH, CE)
        ops.queue_dw(mv(dQ1), mv(e), (gW1, CZ + CE, EH), R, EH, CE, db=G[f"{pre}.trunk.0.bias"])
    else:
        ops.side(_grads_1, (dP1, dQ1, e), R)
    ops.linear_dx(mv(dP1), (W1, CZ, EH), mv(de), R, EH, CE, beta=True)
    ops.linear_dx(mv(dQ1), (W1, CZ + CE, EH), mv(de), R, EH, CE, beta=True)
    if not fused:
        ops.linear_dx(mv(dh1), (W1, 0, EH), mv(dz), Pn, EH, CZ, beta=True)          # dz += dh1 W1_z
    _lin_grads(G, f"{pre}.initial_embed.weight", f"{pre}.initial_embed.bias", mv(de), mv(sv["n3"]), R, CE, CS)
    ops.linear_dx(mv(de), mv(P[f"{pre}.initial_embed.weight"]), mv(dn3), R, CE, CS, beta=True)


# --------------------------------------------------------------------------- heads
_HC = {}


def head_const(conf_key=(0.1, 0.1, 20.0, 0.1, 1.5, 1000), device=None):
    """FdHeadConst from the reference constants (residue_constants.py:127-133, 769-781, 819-824).

    conf_key = (coordinate_scaling, min_b, max_b, min_sigma, max_sigma, L[, so3, num_sigma]): a 7th entry that is an
    SO3Diffuser with use_cached_score=True makes the rotation score the bucketised lookup in its score_norms table
    (so3_diffuser.py:293-299), uploaded once per device."""
    so3 = conf_key[6] if len(conf_key) > 6 else None
    cached = so3 is not None and device is not None
    conf_key = tuple(conf_key[:6]) + ((str(device),) if cached else ())
    # the table-carrying constants live ON the diffuser (a cache keyed by id(so3) could hand a new diffuser the stale
    # device table of a garbage-collected one); the table-free ones are keyed by their six scalars
    store = so3.__dict__.setdefault("_fd_head_const", {}) if cached else _HC
    if conf_key not in store:
        cs, min_b, max_b, min_s, max_s, L = conf_key[:6]
        n = np.array([-0.525, 1.363, 0.000]); ca = np.zeros(3); c = np.array([1.526, -0.000, -0.000])
        cb = np.array([-0.529, -0.774, -1.205]); o = np.array([0.627, 1.062, 0.000])
        ex = c - ca
        ey = ca - n
        exn = ex / np.linalg.norm(ex)
        eyn = ey - np.dot(ey, exn) * exn
        eyn = eyn / np.linalg.norm(eyn)
        ez = np.cross(exn, eyn)
        Rd = np.stack([exn, eyn, ez], 1)
        hc = hip.FdHeadConst()
        for i, v in enumerate(np.concatenate([n, ca, c, cb, o]).astype(np.float32)):
            hc.atoms[i] = float(v)
        for i, v in enumerate(Rd.astype(np.float32).reshape(-1)):
            hc.Rd[i] = float(v)
        for i, v in enumerate(c.astype(np.float32)):
            hc.td[i] = float(v)
        hc.coord_scale = cs
        hc.exp_max_sigma = float(np.exp(max_s))
        hc.exp_min_sigma = float(np.exp(min_s))
        hc.min_b, hc.max_b, hc.L = min_b, max_b, L
        if cached:
            tab, om = so3.device_score_norms(device), so3.device_tables(device)[1]
            assert tab.shape[1] == om.numel()
            hc.score_norms, hc.omega_grid, hc.n_omega = tab.data_ptr(), om.data_ptr(), om.numel()
            hc._tables = (tab, om)            # keeps the device buffers alive as long as the struct
        store[conf_key] = hc
    return store[conf_key]


_SG = {}


def head_tables(dconf, device):
    """(FdHeadConst, sigma grid) of a diffuser configuration on `device`."""
    num_sigma = dconf[7] if len(dconf) > 7 else 1000
    return head_const(dconf, device), sigma_grid(device, dconf[3], dconf[4], num_sigma)


def sigma_grid(device, min_sigma=0.1, max_sigma=1.5, num_sigma=1000):
    """so3_diffuser.py:182-186 discrete_sigma (float64, numpy op sequence) on the device."""
    key = (str(device), min_sigma, max_sigma, num_sigma)
    if key not in _SG:
        t = np.linspace(0.0, 1.0, num_sigma)
        g = np.log(t * np.exp(max_sigma) + (1 - t) * np.exp(min_sigma))
        _SG[key] = torch.tensor(g, dtype=torch.float64, device=device)
    return _SG[key]


def heads_fwd(P, node, quat, trans, feats, B, N, dconf, sc_ca_out=None):
    """Score heads, psi head, backbone atoms (ipa_pytorch.py:650-672, score_network.py:199-214)."""
    tp = "score_model.torsion_pred"
    dev = node
    R = B * N
    h1 = empty((R, CS), dev); h2 = empty((R, CS), dev); u = empty((R, 2), dev)
    ops.linear(mv(node), mv(P[f"{tp}.linear_1.weight"]), P[f"{tp}.linear_1.bias"], mv(h1), R, CS, CS, relu=True)
    ops.linear(mv(h1), mv(P[f"{tp}.linear_2.weight"]), P[f"{tp}.linear_2.bias"], mv(h2), R, CS, CS, resid=mv(node))
    ops.linear(mv(h2), mv(P[f"{tp}.linear_final.weight"]), P[f"{tp}.linear_final.bias"], mv(u), R, 2, CS)
    hc, sg = head_tables(dconf, dev.device)
    rot = empty((B, N, 3), dev, torch.float64); ts = empty((B, N, 3), dev); rig = empty((B, N, 7), dev)
    psi = empty((B, N, 2), dev); a37 = empty((B, N, 37, 3), dev); a14 = empty((B, N, 14, 3), dev)
    gt = feats["torsion_angles_sin_cos"]
    tt = feats["t"].float().contiguous()
    lib().call("fd_heads_fwd", feats["rigids_t"], quat, trans, u, (gt, 4), 14, feats["fixed_mask"], feats["res_mask"],
               tt, sg, sg.numel(), hc, rot, ts, rig, psi, a37, a14, sc_ca_out, B, N)
    out = dict(psi=psi, rot_score=rot, trans_score=ts, rigids=rig, atom37=a37, atom14=a14)
    return out, dict(node=node, h1=h1, h2=h2, u=u, quat=quat, trans=trans, psi=psi, t=tt, hc=hc, sg=sg, B=B, N=N)


def heads_bwd(P, G, sv, feats, d_out, dnode):
    """d_out: dict of optional grads for psi/rot_score/trans_score/rigids/atom37.  Returns (dquat, dtrans)
    of the final frame; accumulates dnode (+=)."""
    tp = "score_model.torsion_pred"
    B, N = sv["B"], sv["N"]
    R = B * N
    dev = sv["node"]
    dq = empty((R, 4), dev); dt = empty((R, 3), dev); du = empty((R, 2), dev)

    def g(k, dtype=torch.float32):
        v = d_out.get(k)
        return None if v is None else v.to(dtype).contiguous()

    lib().call("fd_heads_bwd", feats["rigids_t"], sv["quat"], sv["trans"], sv["u"], sv["psi"], feats["fixed_mask"],
               feats["res_mask"], sv["t"], sv["sg"], sv["sg"].numel(), sv["hc"], g("rot_score", torch.float64),
               g("trans_score"), g("rigids"), g("psi"), g("atom37"), dq, dt, du, B, N)
    _lin_grads(G, f"{tp}.linear_final.weight", f"{tp}.linear_final.bias", mv(du), mv(sv["h2"]), R, 2, CS)
    dh2 = empty((R, CS), dev)
    ops.linear_dx(mv(du), mv(P[f"{tp}.linear_final.weight"]), mv(dh2), R, 2, CS)
    ops.add_view(mv(dnode), mv(dh2), R, CS)                                   # residual branch
    _lin_grads(G, f"{tp}.linear_2.weight", f"{tp}.linear_2.bias", mv(dh2), mv(sv["h1"]), R, CS, CS)
    dh1 = empty((R, CS), dev)
    ops.linear_dx(mv(dh2), mv(P[f"{tp}.linear_2.weight"]), mv(dh1), R, CS, CS, gate=mv(sv["h1"]))
    _lin_grads(G, f"{tp}.linear_1.weight", f"{tp}.linear_1.bias", mv(dh1), mv(sv["node"]), R, CS, CS)
    ops.linear_dx(mv(dh1), mv(P[f"{tp}.linear_1.weight"]), mv(dnode), R, CS, CS, beta=True)
    return dq, dt


# --------------------------------------------------------------------------- whole network
def _prep_feats(feats):
    """Cast / lay out the reference input dict (score_network.py:183-193) for the kernels."""
    f = {}
    f["res_mask"] = feats["res_mask"].to(torch.float32).contiguous()
    f["fixed_mask"] = feats["fixed_mask"].to(torch.float32).contiguous()
    f["seq_idx"] = feats["seq_idx"].to(torch.int64).contiguous()
    f["t"] = feats["t"].contiguous()
    f["sc_ca_t"] = feats["sc_ca_t"].to(torch.float32).contiguous()
    f["rigids_t"] = feats["rigids_t"].to(torch.float32).contiguous()
    f["torsion_angles_sin_cos"] = feats["torsion_angles_sin_cos"].to(torch.float32).contiguous()
    return f


def _cached(cache, key, fn):
    """Mask- / weight-derived constants of a sampling run (the sampler owns `cache` for one trajectory: masks and
    weights do not change between its 500 forwards, so ~15 tiny launches per forward are computed once)."""
    if cache is None:
        return fn()
    if key not in cache:
        cache[key] = fn()
    return cache[key]


def _tfmr_layers(P, b):
    """model.ipa.seq_tfmr_num_layers, read off the state_dict (ipa_pytorch.py:584-593: nn.TransformerEncoder(layer, num_layers))"""
    n = 0
    while f"score_model.trunk.seq_tfmr_{b}.layers.{n}.linear1.weight" in P:
        n += 1
    return n


def _fork_bb_update(P, b, n3, dmask, quat, trans, R, box, like):
    """the backbone update beside the edge transition (sampling): the tensors the branch reads are bound HERE, not looked up
    through the caller's loop variables when the side stream runs (ops.fork), and it keeps its own profiling range"""
    def run(n3=n3, quat=quat, trans=trans):
        with rng(f"node_transition_{b}.fwd"):
            box.update(r=bb_update_fwd(P, b, n3, dmask.view(-1), quat, trans, R))
    ops.fork(run, like, keep=(n3, quat, trans, dmask))


def forward(P, feats, num_blocks, dconf=(0.1, 0.1, 20.0, 0.1, 1.5, 1000), tfmr_bool_mask=False, save=True, cache=None,
            sc_ca_out=None):
    """ScoreNetwork.forward.  Returns (outputs, saved-for-backward or None).  `cache`: see _cached (no-grad only)."""
    if save:
        cache = None
    # training: the weight operand of every node-level GEMM of this step, pre-split once (ops.weight_planes; None = not a flat buffer)
    planes = ops.weight_planes(P) if save else None
    if not save:
        ops.set_weight_planes(None)
    f = _prep_feats(feats)
    B, N = f["res_mask"].shape
    R = B * N
    L = lib()
    mask = f["res_mask"]
    dev = mask
    if not (mask.is_cuda and torch.cuda.is_current_stream_capturing()):
        ops.edge_sched_init(mask.device)      # (the tile-counter pool of the fused edge kernels: never allocated inside a capture)
    if cache is not None:
        # the cache is only valid for the static mask buffers it was built from
        sig = (B, N, mask.data_ptr(), f["fixed_mask"].data_ptr(), bool(tfmr_bool_mask))
        if cache.get("_sig") != sig:
            cache.clear()
            cache["_sig"] = sig
    # frames and the embedder's timestep argument in one launch (instead of a slice copy, a scaled slice copy and t * 1e4)
    rig = f["rigids_t"]
    quat = empty((R, 4), dev); trans = empty((R, 3), dev)                      # trans: scale_rigids (A -> nm)
    t32 = f["t"].dtype == torch.float32
    tsc = empty((B,), dev) if t32 else None
    L.call("fd_split_rigids", rig, float(dconf[0]), f["t"] if t32 else None, 10000.0, quat, trans, tsc, R, B)
    f["tscaled"] = tsc if t32 else (f["t"] * 10000).float().contiguous()     # score_network.py:38,43
    with rng("embed.fwd"):
        node0, z, sv_embed, zb = nw.embed_fwd(P, f, B, N, cache, save=save,
                                              zb_next=nw.ipa_w40(P, "score_model.trunk.ipa_0", cache) if num_blocks > 0 else None)
    emask = sv_embed["emask"]

    def _dmask():
        dm = empty((B, N), dev)
        L.call("fd_rowscale", (1 - f["fixed_mask"]).contiguous(), 1, mask.reshape(-1), dm, 1, R, 1)
        return dm
    dmask = _cached(cache, "dmask", _dmask)
    init_node = node0                                                          # already masked (LN rowscale)
    node = node0
    if tfmr_bool_mask:
        key_add = _cached(cache, "key_add", lambda: torch.where(mask > 0, torch.zeros_like(mask),
                                                                 torch.full_like(mask, float("-inf"))))
    else:
        key_add = (1 - mask).contiguous()
    stages = []
    # sampling: the four skip_embed products depend on nothing but the embedder's output -- ONE GEMM per forward against the
    # zero-padded row-concatenation of their weights (folded once per trajectory) writes the skip columns of every block's
    # [LayerNorm input | skip] buffer; block b's buffer is the column range [320 b, 320 b + 320) of one [R, 320 nb] tensor
    # (its first 256 columns are written later, by the block's IPA linear_out)
    cat_all = None
    if (cache is not None and not save and num_blocks > 1 and opts.merge_skip_embed and not ops.fork_ok(mask)
            and all(_tfmr_layers(P, b) >= 1 for b in range(num_blocks))
            and ops.ln_linear_ok((node, 0, TD), (node, 0, TD), R, 3 * TD, TD)):
        tp = "score_model.trunk"
        if "skip_all" not in cache:
            Wp = torch.zeros((num_blocks * TD, CS), device=mask.device)
            bp = torch.zeros((num_blocks * TD,), device=mask.device)
            for b in range(num_blocks):
                Wp[b * TD + CS:(b + 1) * TD] = P[f"{tp}.skip_embed_{b}.weight"]
                bp[b * TD + CS:(b + 1) * TD] = P[f"{tp}.skip_embed_{b}.bias"]
            cache["skip_all"] = (Wp, bp)
        Wp, bp = cache["skip_all"]
        cat_all = empty((R, num_blocks * TD), dev)
        ops.linear(mv(init_node), mv(Wp), bp, mv(cat_all), R, num_blocks * TD, CS)
    for b in range(num_blocks):
        pre = f"score_model.trunk.ipa_{b}"
        nl = _tfmr_layers(P, b)
        # sampling (ops.ln_linear_ok): LayerNorms run inside the launches of their GEMM consumers.  ipa_ln: linear_out and
        # skip_embed write the two column ranges of one [R, 320] buffer and the first transformer layer's in_proj normalises
        # the first 256 itself; the transition's LayerNorm: inside the edge transition's per-residue GEMM (blocks that have one)
        fold1 = (not save) and nl >= 1 and node.is_cuda == z.is_cuda and ops.ln_linear_ok((node, 0, TD), (node, 0, TD), R, 3 * TD, TD)
        fold6 = fold1 and b < num_blocks - 1 and et_folds_node_terms(cache, save)
        # cat: matrix view (tensor, column offset, row stride) of the block's [LayerNorm input | skip] buffer
        cat = ((cat_all, b * TD, num_blocks * TD) if cat_all is not None else (empty((R, TD), dev), 0, TD)) if fold1 else None
        ops.join(node)          # the backbone update of the block in front (sampling: a second graph branch)
        if fold1 and cat_all is None:
            # skip_embed depends on nothing but the embedder's output: beside the IPA launches (its own columns of `cat`)
            tp = "score_model.trunk"
            ops.fork(lambda b=b, cat=cat: ops.linear(mv(init_node), mv(P[f"{tp}.skip_embed_{b}.weight"]),
                                                     P[f"{tp}.skip_embed_{b}.bias"], (cat[0], cat[1] + CS, cat[2]), R, 64, CS), node,
                     keep=(init_node, cat[0]))
        with rng(f"ipa_{b}.fwd"):
            x1, sv_ipa = nw.ipa_fwd(P, pre, mv(node), z, quat, trans, mask.view(-1), B, N, cache, zb=zb,
                                    out_view=cat if fold1 else None, save=save)
        with rng(f"seq_tfmr_{b}.fwd"):
            pend_ln = None      # a LayerNorm whose launch is folded into its consumer's (nw.tfmr_layer_fwd)
            if fold1:
                tp = "score_model.trunk"
                ops.join(node)
                u = u0 = empty((R, TD), dev)             # written by the first layer's in_proj launch
                pend_ln = (cat, P[f"{tp}.ipa_ln_{b}.weight"], P[f"{tp}.ipa_ln_{b}.bias"], None, CS, u0)
                sv_ln = None
            else:
                u, sv_ln = nw.ln_skip_fwd(P, b, x1, init_node, R)
                u0 = u
            sv_t = []
            for l in range(nl):
                # eval + no_grad fast path of nn.TransformerEncoder: padded rows of its output are zeroed (nested-tensor
                # round trip) -- the row mask rides on the last layer's LayerNorm instead of a launch of its own
                u, s_, pend_ln = nw.tfmr_layer_fwd(P, f"score_model.trunk.seq_tfmr_{b}.layers.{l}", u, key_add, B, N, save=save,
                                                   out_rowscale=mask.view(-1) if (tfmr_bool_mask and l == nl - 1) else None,
                                                   x_ln=pend_ln, defer_out_ln=not save)
                sv_t.append(s_)
        with rng(f"node_transition_{b}.fwd"):
            n3, sv_pn = nw.post_node_fwd(P, b, u, u0, mask.view(-1), R, u2_ln=pend_ln, defer_ln=fold6)
            pend_ln = None
        sv_et = None
        if fold6:
            n3_ln, sv_pn = sv_pn, None
            n3 = n3_ln[5]
            # the backbone update needs the block's node output only: beside the edge transition, joined by the next block
            bbo = {}
            with rng(f"edge_transition_{b}.fwd"):
                z, sv_et, zb = edge_transition_fwd(
                    P, b, None, z, emask, B, N, save=save, cache=cache,
                    zb_next=nw.ipa_w40(P, f"score_model.trunk.ipa_{b + 1}", cache), n3_ln=n3_ln,
                    after_terms=lambda b=b, n3=n3, quat=quat, trans=trans: _fork_bb_update(P, b, n3, dmask, quat, trans, R, bbo, node))
            q2, t2, sv_bb = bbo["r"]
        else:
            with rng(f"node_transition_{b}.fwd"):
                q2, t2, sv_bb = bb_update_fwd(P, b, n3, dmask.view(-1), quat, trans, R)
            if b < num_blocks - 1:
                with rng(f"edge_transition_{b}.fwd"):
                    z, sv_et, zb = edge_transition_fwd(P, b, n3, z, emask, B, N, save=save, cache=cache,
                                                       zb_next=nw.ipa_w40(P, f"score_model.trunk.ipa_{b + 1}", cache))
        stages.append(dict(ipa=sv_ipa, ln=sv_ln, tfmr=sv_t, pn=sv_pn, bb=sv_bb, et=sv_et))
        node, quat, trans = n3, q2, t2
        if not save:
            stages[-1] = None
    ops.join(node)
    with rng("heads.fwd"):
        # sampling: the heads also write the predicted CA positions where the loop keeps its self-conditioning input (sc_ca_out,
        # handed down by sampler.sample through the module): no copy launch between the forward and the reverse step
        sc_out = sc_ca_out if not save else None
        assert sc_out is None or (sc_out.is_contiguous() and sc_out.dtype == torch.float32 and tuple(sc_out.shape) == (B, N, 3))
        out, sv_h = heads_fwd(P, node, quat, trans, f, B, N, dconf, sc_ca_out=sc_out)
    if not save:
        return out, None
    return out, dict(feats=f, embed=sv_embed, stages=stages, heads=sv_h, B=B, N=N, num_blocks=num_blocks,
                     bool_mask=tfmr_bool_mask, mask=mask, dmask=dmask, planes=planes)


def backward(P, G, sv, d_out, on_done=None):
    """Gradients of sum_k <d_out[k], out[k]> w.r.t. every parameter, accumulated into G.
    on_done(tag): called when every gradient launch of a parameter group has been ISSUED -- "heads", then block
    nb-1 ... 0, then "embed" -- so a data-parallel caller can start that group's all-reduce under the rest of the
    backward pass (dist.OverlapAllReduce)."""
    notify = on_done if on_done is not None else (lambda tag: None)
    hooked = on_done is not None
    B, N, nb = sv["B"], sv["N"], sv["num_blocks"]
    R, Pn = B * N, B * N * N
    f = sv["feats"]
    dev = sv["mask"]
    if sv["bool_mask"]:
        raise NotImplementedError("backward is defined for the training-mode (additive) transformer mask")
    # every zero-initialised accumulator of the pass from one memset (per block: the node-term sums of the edge transition
    # [R,2*(128+384)], du0, ds, dframe, ...: ~R * 1,650 floats; IPA's dproj [R,6816] is assigned, not accumulated)
    ops.reset_dw_queue()        # nothing an interrupted earlier pass queued may leak into this one's gradients
    ops.set_weight_planes(sv.get("planes"))     # the forward's split of the parameter buffer (the weights have not changed since)
    try:
        with ops.zero_arena(R * (nb * 2048 + 1024) + 65536 if opts.zero_arena else 0, dev):
            _backward(P, G, sv, d_out, notify, hooked)
    except BaseException:
        ops.reset_dw_queue()
        raise
    finally:
        ops.set_weight_planes(None)


def _backward(P, G, sv, d_out, notify, hooked=False):
    B, N, nb = sv["B"], sv["N"], sv["num_blocks"]
    R, Pn = B * N, B * N * N
    f = sv["feats"]
    dev = sv["mask"]
    dnode = zeros((R, CS), dev)
    nw.embed_regen_early(sv["embed"], G)
    with rng("heads.bwd"):
        dq, dt = heads_bwd(P, G, sv["heads"], f, d_out, dnode)
    ops.flush_dw()
    notify("heads")
    dinit = zeros((R, CS), dev)
    dz = None
    # (a data-parallel caller's on_done hook needs every gradient launch of a group ISSUED when it fires: no deferral then)
    defer_dw = bool(opts.defer_node_dw and not hooked)
    pend = None     # (dzb, W40) of the IPA block just processed whose term dz += dzb W40 the next consumer still has to add
    for b in reversed(range(nb)):
        st = sv["stages"][b]
        dn3 = dnode
        dz_in = None
        if st["et"] is not None:
            dz_in = empty((Pn, CZ), dev)
            with rng(f"edge_transition_{b}.bwd"):
                edge_transition_bwd(P, G, b, st["et"], dz, dz_in, dn3, dzb_next=pend, flush_behind_launch=defer_dw)
            pend = None
        dframe = zeros((R, 12), dev)
        # IPA backward needs dx1, which needs dn3 complete (incl. bb_update's contribution), but bb_update's
        # input-frame gradient needs the IPA's dframe: split bb_update in two steps via a zero dframe first.
        with rng(f"node_transition_{b}.bwd"):
            dq_in, dt_in = bb_update_bwd(P, G, b, st["bb"], dq, dt, None, dn3)
            du0 = zeros((R, TD), dev)
            du2 = nw.post_node_bwd(P, G, b, st["pn"], dn3, du0)
        with rng(f"seq_tfmr_{b}.bwd"):
            du = du2
            for l in reversed(range(len(st["tfmr"]))):
                du = nw.tfmr_layer_bwd(P, G, f"score_model.trunk.seq_tfmr_{b}.layers.{l}", st["tfmr"][l], du)
            ops.add_view(mv(du0), mv(du), R, TD)
            dx1 = empty((R, CS), dev)
            nw.ln_skip_bwd(P, G, b, st["ln"], du0, dx1, dinit)
        ds = zeros((R, CS), dev)
        # the z gradient of this IPA (dzb W40) is added by the consumer of dz when that is a fused edge-transition backward
        # (the transition of block b - 1: its kernel's prologue); else here (fd_ipa_dz_acc)
        defer = b > 0 and fused_edge() and opts.fused_ln_bwd and sv["stages"][b - 1]["et"] is not None
        dz_acc = dz_in is not None
        if dz_in is None and not defer:
            # last block: no edge transition behind this IPA, its z gradient IS dz (ipa_bwd assigns)
            dz_in = empty((Pn, CZ), dev)
        with rng(f"ipa_{b}.bwd"):
            pend = nw.ipa_bwd(P, G, f"score_model.trunk.ipa_{b}", st["ipa"], dx1, mv(ds), dz_in, dframe, dz_accumulate=dz_acc,
                              defer_dz=defer)
        # fold the IPA frame gradients (dL/dR, dL/dt of the block's input frame) into (dq, dt)
        _frame_grad_fold(st["bb"]["quat"], dframe, dq_in, dt_in, R)
        dq, dt, dnode, dz = dq_in, dt_in, ds, dz_in
        # every node-level weight gradient of the block in one launch on the side stream: now -- or (defer_dw) behind the fused
        # backward of the NEXT edge transition, i.e. beside that block's node-level phase instead of in front of its full-chip kernel
        if not (defer_dw and b > 0 and sv["stages"][b - 1]["et"] is not None):
            ops.flush_dw()
        notify(b)
    # node = init_node at block 0 input; both carry gradient into the node embedder
    ops.add_view(mv(dnode), mv(dinit), R, CS)
    with rng("embed.bwd"):
        nw.embed_bwd(P, G, sv["embed"], dnode, dz)
    ops.flush_dw()
    notify("embed")
    ops.join_grad_stream()   # the node-level weight gradients ran beside the dX chain (ops.side)


def _frame_grad_fold(quat, dframe, dq, dt, R):
    """dq += d(dL/dR)/dq, dt += dL/dt, via fd_bb_update_bwd with zero upstream (identity update)."""
    dev = quat
    z4 = zeros((R, 4), dev); z3 = zeros((R, 3), dev); z6 = zeros((R, 6), dev); zm = zeros((R,), dev)
    dq2 = empty((R, 4), dev); dt2 = empty((R, 3), dev); j1 = empty((R, 6), dev); j2 = empty((R, 6), dev)
    # with dmask = 0, upd = 0 and zero upstream grads the kernel returns exactly the dframe contribution
    lib().call("fd_bb_update_bwd", z4, z3, dframe, zm, z6, quat, dq2, dt2, j1, j2, R)
    ops.add_view(mv(dq), mv(dq2), R, 4)
    ops.add_view(mv(dt), mv(dt2), R, 3)
