"""fd_gemm tiles 12 / 13 / 14 (activations x PRE-SPLIT weights, csrc/fd_gemm_w.h) and fd_split_planes against float64:
y = x W^T (weights k-contiguous) and dx = dy W (weights row-contiguous, LDS transpose read), ragged M / N, the full
epilogue, split-K, weights addressed as a sub-matrix of the split buffer (offset + row stride).  CPU tier: the kernel
source under the SIMT interpreter; GPU tier: the gfx950 build at the node-level shapes of the training step.
Reference: nn.Linear forward / input gradient, model/ipa_pytorch.py:101-166."""
import numpy as np
import pytest
import torch


def planes_of(lib, buf):
    """[3, n] uint16 planes of a flat fp32 buffer (fd_split_planes)."""
    n = buf.numel()
    assert n % 8 == 0
    pl = torch.empty((3, n), dtype=torch.int16, device=buf.device)
    lib.call("fd_split_planes", buf, n, pl)
    return pl


def test_split_planes_exact_emu(emu_lib):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4096, generator=g) * torch.exp(4 * torch.randn(4096, generator=g))
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, 1e-30, 1.17549435e-38, -2.5])
    pl = planes_of(emu_lib, x)
    t = (pl.to(torch.int32) << 16).view(torch.float32).double()          # bf16 bit patterns -> values
    s = t[0] + t[1] + t[2]
    # three round-to-nearest bf16 terms cover the 24-bit significand: exact (denormal residuals excepted)
    big = x.abs() > 1e-30
    assert torch.equal(s[big].float(), x[big])
    assert float((s - x.double()).abs().max()) < 1e-37


def _run(lib, dev, M, N, K, b_kc, tile, epi=False, ksplit=1, seed=0, sub=False):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g)
    if b_kc:                                  # y = x W^T: W [N, K]
        Wn, Wk = N, K
    else:                                     # dx = dy W: W [K, N]
        Wn, Wk = K, N
    # the weight sits inside a larger "flat parameter buffer" at an offset, optionally as a column block of a wider matrix
    ld = Wk + (24 if sub else 0)
    off = 64 + (8 if sub else 0)
    flat = torch.randn(off + Wn * ld + 64, generator=g)
    W = flat[off:off + Wn * ld].view(Wn, ld)[:, :Wk]
    Bmat = W.t() if b_kc else W               # [K, N]
    ref = 0.5 * (A.double() @ Bmat.double())
    C = torch.full((M, N), 7.0)
    kw = {}
    if epi:
        bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g); gate = torch.randn(M, N, generator=g)
        rows = torch.rand(M, generator=g)
        ref = torch.clamp(ref + bias.double(), min=0)
        ref = torch.where(gate > 0, ref, torch.zeros_like(ref)) * rows.double()[:, None]
        ref = ref + resid.double() + 7.0
        kw = dict(bias=bias.to(dev), resid=resid.to(dev), ld_resid=N, gate=gate.to(dev), ld_gate=N, rowscale=rows.to(dev),
                  relu=True, beta=True)
    if ksplit > 1:
        ref = ref + 7.0                       # split-K adds its partial tiles into C
    flat_d, A_d, C_d = flat.to(dev), A.to(dev), C.to(dev)
    pl = planes_of(lib, flat_d)
    b_str = (1, ld) if b_kc else (ld, 1)
    lib.gemm(A_d, flat_d, C_d, M, N, K, (K, 1), b_str, N, b_off=off, alpha=0.5, tile=tile, ksplit=ksplit,
             b_planes=(pl.data_ptr() + 2 * off, flat.numel()), **kw)
    if dev != "cpu":
        torch.cuda.synchronize()
    err = (C_d.cpu().double() - ref).abs().max().item()
    return err / (ref.abs().max().item() + 1e-9)


@pytest.mark.parametrize("tile", [12, 13, 14])
@pytest.mark.parametrize("b_kc", [True, False])
def test_gemm_w_emu(emu_lib, tile, b_kc):
    # ragged M (rows clamped), N a multiple of 8 but not of the tile, K a multiple of 16; 1..7 stages (the ring's prologue cases)
    for (M, N, K) in ((70, 72, 16), (130, 136, 48), (64, 128, 112)):
        assert _run(emu_lib, "cpu", M, N, K, b_kc, tile) < 2e-6, (M, N, K)
    assert _run(emu_lib, "cpu", 100, 72, 64, b_kc, tile, epi=True) < 2e-6
    assert _run(emu_lib, "cpu", 100, 72, 64, b_kc, tile, sub=True) < 2e-6
    assert _run(emu_lib, "cpu", 70, 40, 160, b_kc, tile, ksplit=3) < 2e-6


def test_gemm_w_plan_emu(emu_lib):
    """auto selection: planes + a node-level shape -> tiles 12-14; without planes, batched, or K % 16 != 0 -> the old tiles"""
    from se3_diffusion_amd import hip
    import ctypes
    x = torch.zeros(8)
    d = hip.FdGemmDesc()
    d.A = d.B = d.C = x.data_ptr()
    d.M, d.N, d.K = 3840, 320, 320
    d.a_rs, d.a_cs, d.b_rs, d.b_cs, d.ldc = 320, 1, 1, 320, 320
    d.alpha = 1.0
    assert emu_lib.cdll.fd_gemm_plan(ctypes.byref(d)) == 10
    d.b_planes, d.b_plane_stride = x.data_ptr(), 1024
    assert emu_lib.cdll.fd_gemm_plan(ctypes.byref(d)) in (12, 13, 14)
    d.N = 6816
    assert emu_lib.cdll.fd_gemm_plan(ctypes.byref(d)) == 12
    d.K = 328
    assert emu_lib.cdll.fd_gemm_plan(ctypes.byref(d)) not in (12, 13, 14)
    d.K, d.batch = 320, 4
    assert emu_lib.cdll.fd_gemm_plan(ctypes.byref(d)) not in (12, 13, 14)


@pytest.mark.gpu
def test_gemm_w_gpu(hip_lib):
    import parity_log
    worst = 0.0
    for tile in (12, 13, 14):
        for (M, N, K) in ((3840, 320, 320), (3840, 6816, 256), (3840, 256, 2688), (1000, 264, 6816), (3840, 960, 320)):
            for b_kc in (True, False):
                worst = max(worst, _run(hip_lib, "cuda", M, N, K, b_kc, tile))
        worst = max(worst, _run(hip_lib, "cuda", 3840, 256, 256, True, tile, epi=True))
        worst = max(worst, _run(hip_lib, "cuda", 3840, 320, 320, False, tile, epi=True, sub=True))
        worst = max(worst, _run(hip_lib, "cuda", 3840, 256, 6816, False, tile, ksplit=8))
    with parity_log.case("gemm_w_tiles_12_13_14_vs_fp64"):
        parity_log.out("C", worst)
    assert worst < 5e-6, worst      # (K = 6816: fp32 accumulation over 6,816 terms, measured 2.6e-6 of the maximum)
