"""Microbenchmark: fd_node_chain (transition: 256 x 3 layers; transformer feed-forward: 320 x 2 layers) against the launch
sequences it replaces (fd_gemm per layer + fd_layernorm), forward and backward, at the row counts of sampling and training."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from se3_diffusion_amd import ops  # noqa: E402
import node_chain as nc  # noqa: E402

mv = lambda t: (t, 0, t.shape[-1])


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def graphed(fn):
    """the same launches replayed from a hipGraph (how the sampler runs them)"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                fn()
    return lambda: g.replay()


def main():
    dev = "cuda"
    for rows in (128, 256, 1024, 3840):
        for W, NL in ((256, 3), (320, 2)):
            e = lambda *s: torch.empty(*s, device=dev)
            x = e(rows, W).normal_()
            Ws = [e(W, W).normal_() * 0.08 for _ in range(NL)]
            bs = [e(W).normal_() for _ in range(NL)]
            gm, bt = e(W).normal_(), e(W).normal_()
            out, pre, mn, rs = e(rows, W), e(rows, W), e(rows), e(rows)
            hs = [e(rows, W) for _ in range(NL - 1)]
            img, imgT = nc.node_chain_pack(Ws), nc.node_chain_pack(Ws, backward=True)

            def seq_fwd():
                h = x
                for l in range(NL - 1):
                    ops.linear(mv(h), mv(Ws[l]), bs[l], mv(hs[l]), rows, W, W, relu=True)
                    h = hs[l]
                ops.linear(mv(h), mv(Ws[-1]), bs[-1], mv(pre), rows, W, W, resid=mv(x))
                ops.layernorm(mv(pre), gm, bt, mv(out), rows, W, save=(mn, rs))
            chain_fwd = lambda: nc.node_chain(x, img, out, rows, W, NL, gamma=gm, beta=bt, bias=bs, save=hs, pre=pre, mean=mn, rstd=rs)
            chain_inf = lambda: nc.node_chain(x, img, out, rows, W, NL, gamma=gm, beta=bt, bias=bs)
            chain_fwd()
            dy, dx, dt = e(rows, W).normal_(), e(rows, W), e(rows, W)
            ds = [e(rows, W) for _ in range(NL - 1)]
            dg, db = torch.zeros(W, device=dev), torch.zeros(W, device=dev)

            def seq_bwd():
                ops.layernorm_bwd(mv(dy), mv(pre), gm, mn, rs, mv(dt), rows, W, dgamma=dg, dbeta=db)
                d = dt
                for i, l in enumerate(reversed(range(1, NL))):
                    ops.linear_dx(mv(d), mv(Ws[l]), mv(ds[i]), rows, W, W, gate=mv(hs[l - 1]))
                    d = ds[i]
                ops.linear_dx(mv(d), mv(Ws[0]), mv(dx), rows, W, W, resid=mv(dt))
            chain_bwd = lambda: nc.node_chain(dy, imgT, dx, rows, W, NL, gamma=gm, gate=list(reversed(hs)), save=ds, pre=dt,
                                               ln_in=pre, mean=mn, rstd=rs, dgamma=dg, dbeta=db, backward=True)
            pack = lambda: nc.node_chain_pack(Ws, out=img)
            r = {k: timeit(f) for k, f in dict(seq_fwd=seq_fwd, chain_fwd=chain_fwd, chain_inf=chain_inf, seq_bwd=seq_bwd,
                                               chain_bwd=chain_bwd, pack=pack).items()}
            rg = {k: timeit(graphed(f), reps=5, warm=2) / 10 for k, f in dict(seq_fwd=seq_fwd, chain_inf=chain_inf).items()}
            print(f"rows={rows:5d} W={W} NL={NL}: forward {NL + 1} launches {r['seq_fwd']:6.1f} us | chain(+saves) {r['chain_fwd']:6.1f} | "
                  f"chain(inference) {r['chain_inf']:6.1f} || in a graph: {rg['seq_fwd']:6.1f} vs {rg['chain_inf']:6.1f} || backward "
                  f"{NL + 1} launches {r['seq_bwd']:6.1f} | chain {r['chain_bwd']:6.1f} | pack {r['pack']:5.1f} us", flush=True)


if __name__ == "__main__":
    main()
