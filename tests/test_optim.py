"""FlatAdam (fd_adam_step) against torch.optim.Adam: identical parameter trajectories, aligned views, and checkpoint
state that round-trips between the two optimisers."""
import pytest
import torch

from se3_diffusion_amd.optim import FlatAdam


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(7, 5), (6,), (128, 64), (3,), (33, 17)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]


def _run(dev, steps=4):
    ref = _params(dev, 1)
    mine = _params(dev, 1)
    topt = torch.optim.Adam(ref, lr=1e-2)
    fopt = FlatAdam(mine, lr=1e-2)
    for p in mine:
        assert p.data_ptr() % 64 == 0 and p.grad.data_ptr() % 64 == 0      # 256-byte offsets from the buffer base
    g = torch.Generator().manual_seed(2)
    for _ in range(steps):
        fopt.zero_grad()
        for a, b in zip(ref, mine):
            gr = torch.randn(*a.shape, generator=g).to(dev)
            a.grad = gr.clone()
            b.grad.copy_(gr)                       # gradients are written into the flat buffer's views
        topt.step()
        fopt.step()
    for a, b in zip(ref, mine):
        assert (a.detach() - b.detach()).abs().max() < 2e-6 * (1 + a.detach().abs().max())
    return topt, fopt, ref, mine


def test_flat_adam_emu(use_emu):
    _run("cpu")


def test_flat_adam_checkpoint_roundtrip_emu(use_emu):
    topt, fopt, ref, mine = _run("cpu", steps=2)
    # torch -> flat: continue from torch's state and stay identical
    cont = _params("cpu", 1)
    for c, a in zip(cont, ref):
        c.data.copy_(a.data)
    f2 = FlatAdam(cont, lr=1e-2)
    f2.load_state_dict(topt.state_dict())
    g = torch.Generator().manual_seed(9)
    grads = [torch.randn(*a.shape, generator=g) for a in ref]
    for a, c, gr in zip(ref, cont, grads):
        a.grad = gr.clone()
        c.grad.copy_(gr)
    topt.step()
    f2.step()
    for a, c in zip(ref, cont):
        assert (a.detach() - c.detach()).abs().max() < 2e-6 * (1 + a.detach().abs().max())
    # flat -> torch: the exported state loads into torch.optim.Adam
    t2 = torch.optim.Adam(_params("cpu", 1), lr=1e-2)
    t2.load_state_dict(fopt.state_dict())
    assert float(t2.state_dict()["state"][0]["step"]) == 2.0


@pytest.mark.gpu
def test_flat_adam_gpu(hip_lib):
    _run("cuda", steps=5)


def test_flat_adam_survives_zero_grad_emu(use_emu):
    """nn.Module.zero_grad() (set_to_none=True by default) and fresh autograd gradients between steps: FlatAdam
    re-binds its views, so the step never runs on a stale all-zero flat buffer."""
    ref = _params("cpu", 3)
    mine = _params("cpu", 3)
    topt = torch.optim.Adam(ref, lr=1e-2)
    fopt = FlatAdam(mine, lr=1e-2)
    g = torch.Generator().manual_seed(4)
    for it in range(3):
        grads = [torch.randn(*a.shape, generator=g) for a in ref]
        for b in mine:
            b.grad = None                          # what model.zero_grad() does
        for a, b, gr in zip(ref, mine, grads):
            a.grad = gr.clone()
            b.grad = gr.clone()                    # what autograd does when .grad is None: a fresh tensor
        topt.step()
        fopt.step()
        for b, o in zip(mine, fopt.offsets):       # bound to the flat buffer again
            assert b.grad.data_ptr() == fopt.flat_g.data_ptr() + 4 * o
        fopt.zero_grad()
        assert float(fopt.flat_g.abs().max()) == 0.0
    for a, b in zip(ref, mine):
        assert (a.detach() - b.detach()).abs().max() < 2e-6 * (1 + a.detach().abs().max())


def test_flat_adam_model_zero_grad_only_emu(use_emu):
    """The training loop that zeroes ONLY through nn.Module.zero_grad() (never FlatAdam.zero_grad()): the flat buffer still
    holds the previous step's gradient when autograd hands over fresh tensors -- they must replace it, not be added to it
    (step() used to run on grad_prev + grad_new without an error)."""
    ref = _params("cpu", 5)
    mine = _params("cpu", 5)
    topt = torch.optim.Adam(ref, lr=1e-2)
    fopt = FlatAdam(mine, lr=1e-2)
    g = torch.Generator().manual_seed(6)
    for it in range(4):
        grads = [torch.randn(*a.shape, generator=g) for a in ref]
        for b in mine:
            b.grad = None                          # model.zero_grad(set_to_none=True)
        for a, b, gr in zip(ref, mine, grads):
            a.grad = gr.clone()
            b.grad = gr.clone()                    # autograd: a fresh tensor
        if it == 2:
            fopt.all_reduce_mean()                 # (single process: only the rebind) -- then step() must not fold twice
        topt.step()
        fopt.step()
        for b, o, gr in zip(mine, fopt.offsets, grads):
            assert b.grad.data_ptr() == fopt.flat_g.data_ptr() + 4 * o
            assert torch.equal(b.grad, gr)         # this step's gradient alone
    for a, b in zip(ref, mine):
        assert (a.detach() - b.detach()).abs().max() < 2e-6 * (1 + a.detach().abs().max())


def test_flat_grads_model_zero_grad_only():
    from se3_diffusion_amd.dist import FlatGrads
    ps = _params("cpu", 7)
    fg = FlatGrads(ps)
    g = torch.Generator().manual_seed(8)
    for it in range(3):
        grads = [torch.randn(*a.shape, generator=g) for a in ps]
        for i, (p, gr) in enumerate(zip(ps, grads)):
            p.grad = None if (it == 1 and i == 0) else gr.clone()     # one parameter without a gradient in step 1
        fg.all_reduce_mean()
        for i, (p, gr) in enumerate(zip(ps, grads)):
            want = torch.zeros_like(gr) if (it == 1 and i == 0) else gr
            assert torch.equal(p.grad, want), (it, i)
