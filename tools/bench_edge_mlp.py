"""Microbenchmark: fused edge transition (fd_edge_mlp) vs the unfused launch sequence of trunk.edge_transition_fwd."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import ops, trunk  # noqa: E402
from oracle import framediff_oracle as fo  # noqa: E402  (parameter shapes only)


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="30x128,1x256,8x128,1x128,1x512")
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--fwd-only", action="store_true", help="only the no-save forward (PMC runs)")
    ap.add_argument("--lnb", action="store_true", help="the backward with the fused LayerNorm-backward / dzb W40 prologue "
                    "against the separate kernels")
    a = ap.parse_args()
    dev = "cuda"
    P = {k: v.to(dev) for k, v in fo.synth_params(seed=0, conf=dict(fo.CONF, num_blocks=2)).items()}
    pre = "score_model.trunk.edge_transition_0"
    W1, W2, Wf = P[f"{pre}.trunk.0.weight"], P[f"{pre}.trunk.2.weight"], P[f"{pre}.final_layer.weight"]
    for shp in a.shapes.split(","):
        B, N = (int(x) for x in shp.split("x"))
        R, Pn = B * N, B * N * N
        g = torch.Generator(device=dev).manual_seed(1)
        z = torch.randn(Pn, 128, device=dev, generator=g)
        n3 = torch.randn(R, 256, device=dev, generator=g)
        emask = torch.ones(Pn, device=dev)
        flops = 2.0 * Pn * ops.EDGE_MLP_MACS_PER_ROW
        t_unf = timeit(lambda: trunk.edge_transition_fwd_unfused(P, 0, n3, z, emask, B, N))
        img = ops.edge_mlp_pack(W1, W2, Wf)
        imgT = ops.edge_mlp_pack(W1, W2, Wf, backward=True)
        t_pack = timeit(lambda: ops.edge_mlp_pack(W1, W2, Wf, out=img))
        e = lambda *s: torch.empty(*s, device=dev)
        P1, Q1, Pf, Qf = e(R, 384).normal_(), e(R, 384).normal_(), e(R, 128).normal_(), e(R, 128).normal_()
        out, h1, h2, y, mean, rstd = e(Pn, 128), e(Pn, 384), e(Pn, 384), e(Pn, 128), e(Pn), e(Pn)
        b2, gm, bt = P[f"{pre}.trunk.2.bias"], P[f"{pre}.layer_norm.weight"], P[f"{pre}.layer_norm.bias"]
        kw = dict(p1=P1, q1=Q1, bias2=b2, pf=Pf, qf=Qf, gamma=gm, beta=bt, rowscale=emask, blocks=a.blocks)
        t_inf = timeit(lambda: ops.edge_mlp(z, img, out, Pn, N, **kw))
        if a.fwd_only:
            # + the sampling variant: with the next IPA block's zb as a fourth layer
            W40s, b40s = torch.randn(40, 128, device=dev, generator=g) * 0.1, torch.randn(40, device=dev, generator=g)
            img4s, zbs = ops.edge_mlp_pack(W1, W2, Wf, W40=W40s), e(Pn, 40)
            t_zb = timeit(lambda: ops.edge_mlp(z, img4s, out, Pn, N, zb_out=zbs, zb_bias=b40s, **kw))
            print(f"B={B} N={N}: fused fwd(no save) {t_inf:.3f} ms ({flops / t_inf / 1e9:.0f} TF)   with zb {t_zb:.3f} ms")
            continue
        if a.lnb:
            mh1 = torch.zeros(Pn, 12, dtype=torch.int32, device=dev); mh2 = torch.zeros(Pn, 12, dtype=torch.int32, device=dev)
            ops.edge_mlp(z, img, out, Pn, N, save1=h1, save2=h2, y=y, mean=mean, rstd=rstd, mask1=mh1, mask2=mh2, **kw)
            up, dzb, dy = e(Pn, 128).normal_(), e(Pn, 40).normal_(), e(Pn, 128)
            W40 = e(40, 128).normal_()
            dz, d2, d1, dg, db = e(Pn, 128), e(Pn, 384), e(Pn, 384), torch.zeros(128, device=dev), torch.zeros(128, device=dev)
            mv = lambda t_: (t_, 0, t_.shape[-1])
            L = ops.lib()
            t_acc = timeit(lambda: L.call("fd_ipa_dz_acc", dzb, W40, up, Pn, 1))
            t_ln = timeit(lambda: ops.layernorm_bwd(mv(up), mv(y), gm, mean, rstd, mv(dy), Pn, 128, rowscale=emask, dgamma=dg,
                                                    dbeta=db))
            gk = dict(gmask1=mh2, gmask2=mh1, save1=d2, save2=d1, backward=True, blocks=a.blocks)
            t_b = timeit(lambda: ops.edge_mlp(dy, imgT, dz, Pn, N, **gk))
            lk = dict(ln_y=y, ln_mean=mean, ln_rstd=rstd, ln_gamma=gm, ln_rowscale=emask, dy_out=dy, ln_dgamma=dg, ln_dbeta=db)
            imgB = ops.edge_mlp_pack_bwd(Wf, W2, W1)
            imgZ = ops.edge_mlp_pack_bwd(Wf, W2, W1, W40=W40)
            t_f = timeit(lambda: ops.edge_mlp(up, imgB, dz, Pn, N, **gk, **lk))
            t_fz = timeit(lambda: ops.edge_mlp(up, imgZ, dz, Pn, N, dzb=dzb, **gk, **lk))
            lk2 = dict(lk, ln_dgamma=None, ln_dbeta=None)
            t_fn = timeit(lambda: ops.edge_mlp(up, imgB, dz, Pn, N, **gk, **lk2))
            print(f"B={B} N={N} rows={Pn}: dz_acc {t_acc:.3f} | LN bwd {t_ln:.3f} | backward {t_b:.3f} | sum {t_acc + t_ln + t_b:.3f}"
                  f" || fused LN {t_f:.3f} (vs {t_ln + t_b:.3f}) | fused LN + dzb {t_fz:.3f} | fused LN, no dgamma flush "
                  f"{t_fn:.3f} ms", flush=True)
            continue
        # training forward as the step runs it: saves + packed masks + the next block's zb layer
        mh1 = torch.zeros(Pn, 12, dtype=torch.int32, device=dev); mh2 = torch.zeros(Pn, 12, dtype=torch.int32, device=dev)
        W40, b40, zb = e(40, 128).normal_() * 0.1, e(40).normal_(), e(Pn, 40)
        img4 = ops.edge_mlp_pack(W1, W2, Wf, W40=W40)
        tk = dict(save1=h1, save2=h2, y=y, mean=mean, rstd=rstd, mask1=mh1, mask2=mh2)
        t_trn = timeit(lambda: ops.edge_mlp(z, img, out, Pn, N, **tk, **kw))
        t_trn_zb = timeit(lambda: ops.edge_mlp(z, img4, out, Pn, N, zb_out=zb, zb_bias=b40, **tk, **kw))
        dz, d2, d1 = e(Pn, 128), e(Pn, 384), e(Pn, 384)
        gk = dict(gmask1=mh2, gmask2=mh1, save1=d2, save2=d1, backward=True, blocks=a.blocks)
        t_bwd = timeit(lambda: ops.edge_mlp(y, imgT, dz, Pn, N, **gk))
        up, dzb, dy = e(Pn, 128).normal_(), e(Pn, 40).normal_(), e(Pn, 128)
        dg, db = torch.zeros(128, device=dev), torch.zeros(128, device=dev)
        lk = dict(ln_y=y, ln_mean=mean, ln_rstd=rstd, ln_gamma=gm, ln_rowscale=emask, dy_out=dy, ln_dgamma=dg, ln_dbeta=db)
        imgZ = ops.edge_mlp_pack_bwd(Wf, W2, W1, W40=W40)
        t_bwd_full = timeit(lambda: ops.edge_mlp(up, imgZ, dz, Pn, N, dzb=dzb, **gk, **lk))
        tf = lambda ms: flops / ms / 1e9
        print(f"B={B} N={N} rows={Pn}: unfused fwd {t_unf:.3f} ms ({tf(t_unf):.0f} TF) | fused fwd(no save) {t_inf:.3f} ms "
              f"({tf(t_inf):.0f} TF) | fused fwd(+h1,h2z,y,masks) {t_trn:.3f} ms ({tf(t_trn):.0f} TF) | + zb {t_trn_zb:.3f} ms | "
              f"fused bwd chain(+d2,d1) {t_bwd:.3f} ms ({tf(t_bwd):.0f} TF) | bwd with LN + dzb prologue {t_bwd_full:.3f} ms | "
              f"pack {t_pack * 1e3:.1f} us", flush=True)


if __name__ == "__main__":
    main()
