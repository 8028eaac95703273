"""fd_gemm tile 5 (latency kernel for the node-level GEMMs of sampling: 32x32 tiles, K split over the four waves,
fragments straight from global memory) against float64, every operand layout, ragged M / N, the fused epilogue, the
batched (b, h) form and the automatic selection."""
import pytest
import torch

from test_gemm import _run


LAYOUTS = [(True, True), (True, False), (False, True), (False, False)]
CASES = [(32, 32, 8), (128, 320, 320), (100, 72, 40), (33, 40, 64), (7, 6, 16), (128, 256, 2688), (128, 320, 1280), (50, 72, 968)]


@pytest.mark.parametrize("layout", LAYOUTS)
def test_direct_layouts_emu(emu_lib, layout):
    for (M, N, K) in CASES[:5]:
        assert _run(emu_lib, "cpu", M, N, K, layout[0], layout[1], 5) < 2e-6, (M, N, K)


def test_direct_epilogue_emu(emu_lib):
    assert _run(emu_lib, "cpu", 100, 72, 40, True, True, 5, epi=True) < 2e-6
    assert _run(emu_lib, "cpu", 36, 44, 24, True, False, 5, epi=True) < 2e-6


@pytest.mark.parametrize("layout", LAYOUTS)
def test_direct_long_k_emu(emu_lib, layout):
    """long K on few tiles: the 16 x 16-tile form (K >= 640, K % 16 == 0), ragged M / N, the full epilogue; K % 16 != 0 keeps 32 x 32"""
    assert _run(emu_lib, "cpu", 33, 40, 640, layout[0], layout[1], 5, epi=True, seed=3) < 3e-6
    assert _run(emu_lib, "cpu", 20, 21 if not layout[1] else 36, 1296, layout[0], layout[1], 5, seed=4) < 3e-6
    assert _run(emu_lib, "cpu", 33, 40, 648, layout[0], layout[1], 5, epi=True, seed=5) < 3e-6


def _batched(lib, dev):
    g = torch.Generator().manual_seed(2)
    Bn, H, N, Cc = 2, 3, 20, 24
    q = torch.randn(Bn, N, H * Cc, generator=g)
    kv = torch.randn(Bn, N, H * 2 * Cc, generator=g)
    S = torch.zeros(Bn, H, N, N).to(dev)
    lib.gemm(q.to(dev), kv.to(dev), S, N, N, Cc, (H * Cc, 1), (1, H * 2 * Cc), N, batch=Bn * H, bdiv=H,
             a_bs=(N * H * Cc, Cc), b_bs=(N * H * 2 * Cc, 2 * Cc), c_bs=(H * N * N, N * N), alpha=0.25, tile=5)
    k = kv.view(Bn, N, H, 2 * Cc)[..., :Cc]
    ref = 0.25 * torch.einsum("bihc,bjhc->bhij", q.view(Bn, N, H, Cc).double(), k.double())
    assert (S.cpu().double() - ref).abs().max() < 1e-5


def test_direct_batched_emu(emu_lib):
    _batched(emu_lib, "cpu")


def test_direct_rejects_unsupported(emu_lib):
    from se3_diffusion_amd.hip import FdError
    with pytest.raises(FdError):
        _run(emu_lib, "cpu", 33, 6, 65, True, True, 5)          # K % 8 != 0


def test_direct_is_planned_for_small_problems(emu_lib):
    """auto selection: node-level sampling shapes take tile 5, training shapes do not"""
    import ctypes
    from se3_diffusion_amd.hip import FdGemmDesc

    def plan(M, N, K, batch=1):
        A = torch.zeros(M, K); B = torch.zeros(N, K); C = torch.zeros(M, N)
        d = FdGemmDesc()
        d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
        d.M, d.N, d.K = M, N, K
        d.a_rs, d.a_cs, d.b_rs, d.b_cs, d.ldc = K, 1, 1, K, N
        d.batch, d.bdiv, d.alpha = batch, 1, 1.0
        return emu_lib.cdll.fd_gemm_plan(ctypes.byref(d))

    assert plan(128, 320, 320) == 5
    assert plan(1024, 320, 320) == 5
    assert plan(3840, 320, 320) == 10          # (the 64x64 split-bf16 tile; 2 = its fp32 sibling for K < 256)
    assert plan(3840, 320, 128) == 2
    assert plan(128, 320, 324) != 5


@pytest.mark.gpu
def test_direct_gpu(hip_lib):
    for (a_kc, b_kc) in LAYOUTS:
        for (M, N, K) in CASES:
            assert _run(hip_lib, "cuda", M, N, K, a_kc, b_kc, 5) < 2e-6, (a_kc, b_kc, M, N, K)
        assert _run(hip_lib, "cuda", 100, 72, 40, a_kc, b_kc, 5, epi=True) < 2e-6
    _batched(hip_lib, "cuda")
    assert _run(hip_lib, "cuda", 128, 320, 320, True, True, 0) < 2e-6     # planned onto the latency kernel
