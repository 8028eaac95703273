"""(experiment, GPU only: python tools/experiments/check_node_chain.py)
Node-level residual MLP blocks in one launch (tools/experiments/fd_node_chain.hip) against a float64 restatement of
StructureModuleTransition.forward + node mask (model/ipa_pytorch.py:169-191,644) and of the feed-forward half of a post-norm
TransformerEncoderLayer (built at ipa_pytorch.py:584-595), and of their input-gradient chains (torch.autograd on the float64
restatement).

Tolerance: split-bf16 arithmetic is fp32-accurate -- 5e-6 of the tensor maximum for hidden activations / pre-LayerNorm rows,
2e-5 for LayerNorm outputs and for the gradients."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import node_chain as ops  # noqa: E402


def rel(a, b):
    b = b.detach().double().cpu()
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def _run(dev, rows, W, NL, seed=0, blocks=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x = rn(rows, W)
    Ws = [rn(W, W, sc=0.08) for _ in range(NL)]
    bs = [rn(W, sc=0.3) for _ in range(NL)]
    gamma, beta = 1 + rn(W, sc=0.2), rn(W, sc=0.2)
    mask = (torch.rand(rows, generator=g) > 0.2).float().to(dev)
    dy = rn(rows, W)
    e = lambda *s: torch.empty(*s, device=dev)

    # float64 reference with autograd
    xd = x.double().cpu().requires_grad_(True)
    Wd = [w.double().cpu() for w in Ws]
    bd = [b_.double().cpu() for b_ in bs]
    gd, btd = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    hs = []
    h = xd
    for l in range(NL - 1):
        h = torch.relu(h @ Wd[l].T + bd[l])
        h.retain_grad()
        hs.append(h)
    t = h @ Wd[NL - 1].T + bd[NL - 1] + xd
    t.retain_grad()
    mean = t.mean(-1, keepdim=True)
    var = ((t - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1 / torch.sqrt(var + 1e-5)
    out_ref = (((t - mean) * rstd) * gd + btd) * mask.double().cpu()[:, None]
    out_ref.backward(dy.double().cpu())

    img = ops.node_chain_pack(Ws)
    out, pre, mn, rs = e(rows, W), e(rows, W), e(rows), e(rows)
    saves = [e(rows, W) for _ in range(NL - 1)]
    ops.node_chain(x, img, out, rows, W, NL, gamma=gamma, beta=beta, bias=bs, save=saves, pre=pre, rowscale=mask, mean=mn,
                   rstd=rs, blocks=blocks)
    for l in range(NL - 1):
        assert rel(saves[l], hs[l]) < 5e-6, (l, rel(saves[l], hs[l]))
    assert rel(pre, t) < 5e-6 and rel(out, out_ref) < 2e-5, (rel(pre, t), rel(out, out_ref))
    assert rel(mn, mean[:, 0]) < 5e-6 and rel(rs, rstd[:, 0]) < 2e-5
    # inference form (no saves): bit-identical output
    out2 = e(rows, W)
    ops.node_chain(x, img, out2, rows, W, NL, gamma=gamma, beta=beta, bias=bs, rowscale=mask, blocks=blocks)
    assert torch.equal(out, out2)

    # backward chain on the kernel's own saves
    imgT = ops.node_chain_pack(Ws, backward=True)
    dx, dt = e(rows, W), e(rows, W)
    ds = [e(rows, W) for _ in range(NL - 1)]
    dgm, dbt = torch.zeros(W, device=dev), torch.zeros(W, device=dev)
    ops.node_chain(dy, imgT, dx, rows, W, NL, gamma=gamma, gate=list(reversed(saves)), save=ds, pre=dt, ln_in=pre, rowscale=mask,
                   mean=mn, rstd=rs, dgamma=dgm, dbeta=dbt, backward=True, blocks=blocks)
    flip = sum(float(((saves[l].cpu() > 0) != (hs[l] > 0)).float().mean()) for l in range(NL - 1))
    assert flip < 1e-4
    assert rel(dt, t.grad) < 2e-5, rel(dt, t.grad)
    assert rel(dgm, gd.grad) < 2e-5 and rel(dbt, btd.grad) < 2e-5
    if flip == 0:
        # pre-activation gradients: d(h_l) gated = h_l.grad * [h_l > 0] (the tensors the weight gradients multiply)
        for i, l in enumerate(reversed(range(NL - 1))):
            ref = hs[l].grad * (hs[l] > 0)
            assert rel(ds[i], ref) < 2e-5, (l, rel(ds[i], ref))
        assert rel(dx, xd.grad) < 2e-5, rel(dx, xd.grad)


if __name__ == "__main__":
    for W, NL in ((256, 3), (320, 2)):
        _run("cuda", rows=72, W=W, NL=NL)
        _run("cuda", rows=128, W=W, NL=NL, seed=1)
        _run("cuda", rows=3840, W=W, NL=NL, seed=2)
        _run("cuda", rows=1000, W=W, NL=NL, seed=3, blocks=3)      # few persistent blocks walking several tiles, ragged tail
    print("fd_node_chain: forward and backward agree with the float64 restatement")
