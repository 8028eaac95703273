"""HBM-bound reduction kernels of fd_norm.hip against float64 torch: pair reductions (both code paths), column sums,
LayerNorm forward/backward.  CPU tier = SIMT interpreter; GPU tier = gfx950."""
import pytest
import torch


def _pair_reduce(lib, dev, nb, n, C, which):
    g = torch.Generator().manual_seed(n * 7 + C)
    X = torch.randn(nb, n, n, C, generator=g)
    r0 = torch.randn(nb * n, C, generator=g)
    c0 = torch.randn(nb * n, C, generator=g)
    r, c = r0.clone().to(dev), c0.clone().to(dev)
    lib.call("fd_pair_reduce_acc", X.to(dev), nb, n, C, r if "r" in which else None, c if "c" in which else None, C)
    Xd = X.double()
    if "r" in which:
        ref = r0.double() + Xd.sum(2).reshape(nb * n, C)
        assert (r.cpu().double() - ref).abs().max() < 1e-5 * ref.abs().max()
    if "c" in which:
        ref = c0.double() + Xd.sum(1).reshape(nb * n, C)
        assert (c.cpu().double() - ref).abs().max() < 1e-5 * ref.abs().max()


def test_pair_reduce_emu(emu_lib):
    _pair_reduce(emu_lib, "cpu", 2, 9, 128, "rc")      # fused one-pass kernel, ragged i / j tiles
    _pair_reduce(emu_lib, "cpu", 1, 37, 256, "rc")     # two i chunks
    _pair_reduce(emu_lib, "cpu", 2, 7, 40, "rc")       # C % 128 != 0: two-kernel path
    _pair_reduce(emu_lib, "cpu", 2, 7, 128, "r")
    _pair_reduce(emu_lib, "cpu", 2, 7, 128, "c")


@pytest.mark.gpu
def test_pair_reduce_gpu(hip_lib):
    _pair_reduce(hip_lib, "cuda", 3, 128, 384, "rc")
    _pair_reduce(hip_lib, "cuda", 1, 200, 128, "rc")   # more than one j tile: atomic row sums
    _pair_reduce(hip_lib, "cuda", 2, 33, 40, "rc")
    _pair_reduce(hip_lib, "cuda", 2, 64, 384, "c")


def _layernorm(lib, dev, rows, C, accum=False, scale=True):
    """fd_layernorm_fwd/bwd (with the fused row mask) against float64 autograd of torch.nn.functional.layer_norm"""
    g = torch.Generator().manual_seed(rows * 3 + C)
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    gamma = torch.randn(C, generator=g)
    beta = torch.randn(C, generator=g)
    rs = (torch.rand(rows, generator=g) > 0.3).float() if scale else None
    dy = torch.randn(rows, C, generator=g)
    dx0 = torch.randn(rows, C, generator=g)
    dg0, db0 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = torch.empty(rows, C).to(dev)
    mean, rstd = torch.empty(rows).to(dev), torch.empty(rows).to(dev)
    xd = x.to(dev)
    gd = gamma.to(dev)
    rsd = rs.to(dev) if scale else None
    lib.call("fd_layernorm_fwd", xd, C, gd, beta.to(dev), rsd, y, C, mean, rstd, rows, C, 1e-5)
    dx = dx0.clone().to(dev)
    dg, db = dg0.clone().to(dev), db0.clone().to(dev)
    lib.call("fd_layernorm_bwd", dy.to(dev), C, xd, C, gd, rsd, mean, rstd, dx, C, int(accum), dg, db, rows, C)
    X = x.double().requires_grad_(True)
    G, Bt = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(X, (C,), G, Bt, 1e-5)
    if scale:
        ref = ref * rs.double()[:, None]
    ref.backward(dy.double())
    assert (y.cpu().double() - ref.detach()).abs().max() < 2e-5
    want_dx = X.grad + (dx0.double() if accum else 0)
    assert (dx.cpu().double() - want_dx).abs().max() < 5e-5 * max(1.0, float(want_dx.abs().max()))
    assert (dg.cpu().double() - (dg0.double() + G.grad)).abs().max() < 1e-4 * max(1.0, float(G.grad.abs().max()))
    assert (db.cpu().double() - (db0.double() + Bt.grad)).abs().max() < 1e-4 * max(1.0, float(Bt.grad.abs().max()))


def test_layernorm_emu(emu_lib):
    _layernorm(emu_lib, "cpu", 67, 128)                 # C = 128 fast path, ragged row groups
    _layernorm(emu_lib, "cpu", 40, 128, accum=True, scale=False)
    _layernorm(emu_lib, "cpu", 21, 256)                 # node-level path: 4 rows per wave and trip, float4 chunks, ragged last trip
    _layernorm(emu_lib, "cpu", 9, 320, accum=True)      # two chunks per lane, the second on 16 lanes
    _layernorm(emu_lib, "cpu", 70, 512)                 # two full chunks; more than one trip per wave on the interpreter's grid
    _layernorm(emu_lib, "cpu", 5, 64, scale=False)      # one chunk on 16 lanes


@pytest.mark.gpu
def test_layernorm_gpu(hip_lib):
    _layernorm(hip_lib, "cuda", 100003, 128)
    _layernorm(hip_lib, "cuda", 4099, 128, accum=True, scale=False)
    _layernorm(hip_lib, "cuda", 3840, 256)
    _layernorm(hip_lib, "cuda", 3840, 320)              # the training step's node-level shape: one trip per wave
    _layernorm(hip_lib, "cuda", 777, 320, accum=True)
    _layernorm(hip_lib, "cuda", 40000, 512)             # several trips per wave


def _colsum(lib, dev, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    X = torch.randn(rows, C, generator=g)
    o0 = torch.randn(C, generator=g)
    o = o0.clone().to(dev)
    lib.call("fd_colsum_acc", X.to(dev), C, rows, C, o)
    ref = o0.double() + X.double().sum(0)
    assert (o.cpu().double() - ref).abs().max() < 1e-5 * ref.abs().max()


def test_colsum_emu(emu_lib):
    _colsum(emu_lib, "cpu", 300, 128)
    _colsum(emu_lib, "cpu", 77, 30)


@pytest.mark.gpu
def test_colsum_gpu(hip_lib):
    _colsum(hip_lib, "cuda", 100000, 384)
    _colsum(hip_lib, "cuda", 5000, 30)
