"""fd_gemm parity: every operand layout, tile config, tail shape and epilogue
against a float64 numpy contraction.  CPU tier runs the kernel source under the
SIMT interpreter; GPU tier runs the gfx950 build."""
import numpy as np
import pytest
import torch


def _ref(A, B, alpha=1.0):
    return alpha * (A.double() @ B.double())


def _run(lib, dev, M, N, K, a_kc, b_kc, tile, epi=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    if a_kc:
        At = A.contiguous(); a_str = (K, 1)
    else:
        At = A.t().contiguous(); a_str = (1, M)       # stored [K][M]
    if b_kc:
        Bt = B.t().contiguous(); b_str = (1, K)       # stored W[N][K]
    else:
        Bt = B.contiguous(); b_str = (N, 1)
    C = torch.full((M, N), 7.0)
    kw = {}
    ref = _ref(A, B, 0.5)
    if epi:
        bias = torch.randn(N, generator=g)
        resid = torch.randn(M, N, generator=g)
        gate = torch.randn(M, N, generator=g)
        rows = torch.rand(M, generator=g)
        ref = torch.clamp(ref + bias.double(), min=0)
        ref = torch.where(gate > 0, ref, torch.zeros_like(ref)) * rows.double()[:, None]
        ref = ref + resid.double() + 7.0
        kw = dict(bias=bias.to(dev), resid=resid.to(dev), ld_resid=N, gate=gate.to(dev), ld_gate=N,
                  rowscale=rows.to(dev), relu=True, beta=True)
    At, Bt, C = At.to(dev), Bt.to(dev), C.to(dev)
    lib.gemm(At, Bt, C, M, N, K, a_str, b_str, N, alpha=0.5, tile=tile, **kw)
    if dev != "cpu":
        torch.cuda.synchronize()
    err = (C.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-9
    return err / scale


CASES = [
    (64, 64, 32), (128, 128, 64), (100, 72, 40), (33, 6, 65), (130, 40, 128), (256, 384, 96),
]


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
@pytest.mark.parametrize("tile", [1, 2, 3])
def test_gemm_layouts_emu(emu_lib, a_kc, b_kc, tile):
    for (M, N, K) in CASES[:4]:
        assert _run(emu_lib, "cpu", M, N, K, a_kc, b_kc, tile) < 2e-6


def test_gemm_epilogue_emu(emu_lib):
    for tile in (1, 2, 3):
        assert _run(emu_lib, "cpu", 100, 72, 40, True, True, tile, epi=True) < 2e-6
        assert _run(emu_lib, "cpu", 100, 72, 40, False, False, tile, epi=True) < 2e-6


def test_gemm_batched_pair_emu(emu_lib):
    _batched_pair(emu_lib, "cpu")


def _batched_pair(lib, dev):
    g = torch.Generator().manual_seed(1)
    # batched with two-level batch index (b, h) and strided operands, as the IPA q k^T uses
    Bn, H, N, Cc = 2, 3, 20, 24
    q = torch.randn(Bn, N, H * Cc, generator=g)
    kv = torch.randn(Bn, N, H * 2 * Cc, generator=g)
    S = torch.zeros(Bn, H, N, N)
    qd, kvd, Sd = q.to(dev), kv.to(dev), S.to(dev)
    lib.gemm(qd, kvd, Sd, N, N, Cc, (H * Cc, 1), (1, H * 2 * Cc), N, batch=Bn * H, bdiv=H,
             a_bs=(N * H * Cc, Cc), b_bs=(N * H * 2 * Cc, 2 * Cc), c_bs=(H * N * N, N * N), alpha=0.25)
    k = kv.view(Bn, N, H, 2 * Cc)[..., :Cc]
    ref = 0.25 * torch.einsum("bihc,bjhc->bhij", q.view(Bn, N, H, Cc).double(), k.double())
    assert (Sd.cpu().double() - ref).abs().max() < 1e-5
    # pair-broadcast epilogue (edge transition): rows are pairs (b,i,j)
    nres, Cz, Co = 5, 16, 40
    z = torch.randn(Bn * nres * nres, Cz, generator=g)
    W = torch.randn(Co, Cz, generator=g)
    P = torch.randn(Bn * nres, Co, generator=g)
    Q = torch.randn(Bn * nres, Co, generator=g)
    out = torch.zeros(Bn * nres * nres, Co)
    od = out.to(dev)
    lib.gemm(z.to(dev), W.to(dev), od, Bn * nres * nres, Co, Cz, (Cz, 1), (1, Cz), Co,
             pair=(P.to(dev), Q.to(dev), Co, nres), relu=True)
    ref = (z.double() @ W.double().t()).view(Bn, nres, nres, Co) + P.double().view(Bn, nres, 1, Co) \
        + Q.double().view(Bn, 1, nres, Co)
    ref = torch.clamp(ref, min=0).view(-1, Co)
    assert (od.cpu().double() - ref).abs().max() < 1e-5


@pytest.mark.gpu
def test_gemm_gpu(hip_lib):
    for a_kc in (True, False):
        for b_kc in (True, False):
            for tile in (1, 2, 3):
                for (M, N, K) in CASES:
                    assert _run(hip_lib, "cuda", M, N, K, a_kc, b_kc, tile) < 2e-6
                assert _run(hip_lib, "cuda", 100, 72, 40, a_kc, b_kc, tile, epi=True) < 2e-6
    _batched_pair(hip_lib, "cuda")
    assert _run(hip_lib, "cuda", 4096, 384, 384, True, True, 0) < 2e-6


def _mtiles(lib, dev):
    """several M tiles pipelined by one block (persistent loop), with an epilogue and ragged last tile"""
    g = torch.Generator().manual_seed(5)
    for (M, N, K, tile, mt) in [(300, 72, 96, 2, 3), (700, 130, 40, 1, 2), (333, 40, 64, 3, 4)]:
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g)
        bias = torch.randn(N, generator=g)
        C = torch.zeros(M, N).to(dev)
        lib.gemm(A.to(dev), W.to(dev), C, M, N, K, (K, 1), (1, K), N, bias=bias.to(dev), relu=True, tile=tile, mtiles=mt)
        ref = torch.clamp(A.double() @ W.double().t() + bias.double(), min=0)
        assert (C.cpu().double() - ref).abs().max() < 1e-4, (M, N, K)


def test_gemm_mtiles_emu(emu_lib):
    _mtiles(emu_lib, "cpu")


@pytest.mark.gpu
def test_gemm_mtiles_gpu(hip_lib):
    _mtiles(hip_lib, "cuda")


def _splitk(lib, dev):
    g = torch.Generator().manual_seed(3)
    M, N, K = 40, 72, 1000   # dW-like: reduction over many rows, operands row-contiguous (TN)
    dY = torch.randn(K, M, generator=g)
    X = torch.randn(K, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    C = C0.clone().to(dev)
    lib.gemm(dY.to(dev), X.to(dev), C, M, N, K, (1, M), (N, 1), N, ksplit=7)
    ref = C0.double() + dY.double().t() @ X.double()
    assert (C.cpu().double() - ref).abs().max() < 1e-4


def test_gemm_splitk_emu(emu_lib):
    _splitk(emu_lib, "cpu")


@pytest.mark.gpu
def test_gemm_splitk_gpu(hip_lib):
    _splitk(hip_lib, "cuda")


# ---- tile 4: split-bf16 (three exact bf16 terms per fp32 operand, six bf16-MFMA products, fp32 accumulate) ----
SPLIT_LAYOUTS = [(True, True), (True, False), (False, False)]
SPLIT_CASES = [(64, 64, 32), (128, 128, 64), (100, 72, 40), (260, 136, 100), (256, 384, 96)]


@pytest.mark.parametrize("layout", SPLIT_LAYOUTS)
def test_gemm_split_layouts_emu(emu_lib, layout):
    for (M, N, K) in SPLIT_CASES[:4]:
        assert _run(emu_lib, "cpu", M, N, K, layout[0], layout[1], 4) < 2e-6, (M, N, K)


def test_gemm_split_epilogue_emu(emu_lib):
    assert _run(emu_lib, "cpu", 100, 72, 40, True, True, 4, epi=True) < 2e-6
    assert _run(emu_lib, "cpu", 300, 132, 72, True, False, 4, epi=True) < 2e-6


def _persistent(lib, dev, blocks, cases):
    """persistent blocks of the split kernel (several tiles per block, ragged edges, odd stage counts, k tails, fused
    epilogue) == the fresh-block kernel bit for bit (same per-tile arithmetic), and right against fp64"""
    was = lib.cdll.fd_gemm_set_persistent_blocks(blocks)
    try:
        for (M, N, K, b_kc, epi) in cases:
            assert lib.cdll.fd_gemm_set_persistent_blocks(blocks) == blocks
            e1 = _run(lib, dev, M, N, K, True, b_kc, 4, epi=epi, seed=3)
            assert e1 < 2e-6, (M, N, K, e1)
    finally:
        lib.cdll.fd_gemm_set_persistent_blocks(was)


def test_gemm_split_persistent_emu(emu_lib):
    # 3 x 3 and 4 x 2 tiles over 3 persistent blocks: 3, 3, 3 and 3, 3, 2 tiles per block
    _persistent(emu_lib, "cpu", 3, [(700, 300, 40, True, False), (1000, 200, 72, False, True), (520, 260, 16, True, True)])


@pytest.mark.gpu
def test_gemm_split_persistent_gpu(hip_lib):
    _persistent(hip_lib, "cuda", 256, [(256 * 40 + 17, 128 * 13 + 4, 200, True, True), (256 * 70, 1024, 72, False, True)])
    _persistent(hip_lib, "cuda", 5, [(700, 300, 40, True, False), (1000, 200, 72, False, True)])
    # identical to the fresh-block kernel
    g = torch.Generator().manual_seed(9)
    M, N, K = 256 * 64, 1152, 384
    A = torch.randn(M, K, generator=g).cuda(); W = torch.randn(N, K, generator=g).cuda()
    outs = []
    for blocks in (256, 0):
        was = hip_lib.cdll.fd_gemm_set_persistent_blocks(blocks)
        C = torch.zeros(M, N, device="cuda")
        hip_lib.gemm(A, W, C, M, N, K, (K, 1), (1, K), N, tile=4)
        hip_lib.cdll.fd_gemm_set_persistent_blocks(was)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])


def _split_accuracy(lib, dev, M, N, K):
    """the split path carries fp32 accuracy: its error against fp64 is of the size of the fmaf-chain kernel's"""
    g = torch.Generator().manual_seed(9)
    A = (torch.randn(M, K, generator=g) * torch.exp(2 * torch.randn(M, K, generator=g)))   # wide dynamic range
    W = torch.randn(N, K, generator=g)
    ref = A.double() @ W.double().t()
    mag = A.double().abs() @ W.double().abs().t()      # error scale of a K-term fp32 dot product
    errs = {}
    for tile in (1, 4):
        C = torch.zeros(M, N).to(dev)
        lib.gemm(A.to(dev), W.to(dev), C, M, N, K, (K, 1), (1, K), N, tile=tile)
        errs[tile] = float(((C.cpu().double() - ref).abs() / mag).max())
    assert errs[1] < 2e-6 and errs[4] < 2e-6, errs
    assert errs[4] < 3 * errs[1] + 2e-7, errs
    return errs


def test_gemm_split_accuracy_emu(emu_lib):
    _split_accuracy(emu_lib, "cpu", 64, 32, 512)


def test_gemm_split_unsupported_layout_raises(emu_lib):
    from se3_diffusion_amd.hip import FdError
    with pytest.raises(FdError):
        _run(emu_lib, "cpu", 64, 64, 32, False, True, 4)       # A row-contiguous with B k-contiguous: not built
    with pytest.raises(FdError):
        _run(emu_lib, "cpu", 33, 6, 65, True, True, 4)         # unaligned operands


def _splitk4(lib, dev, M=136, N=72, K=1000, ks=5, tile=4):
    g = torch.Generator().manual_seed(4)
    dY = torch.randn(K, M, generator=g)
    X = torch.randn(K, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    C = C0.clone().to(dev)
    lib.gemm(dY.to(dev), X.to(dev), C, M, N, K, (1, M), (N, 1), N, ksplit=ks, tile=tile)
    ref = C0.double() + dY.double().t() @ X.double()
    assert (C.cpu().double() - ref).abs().max() < 2e-6 * ref.abs().max()


def test_gemm_split_splitk_emu(emu_lib):
    _splitk4(emu_lib, "cpu")


@pytest.mark.gpu
def test_gemm_split_gpu(hip_lib):
    for (a_kc, b_kc) in SPLIT_LAYOUTS:
        for (M, N, K) in SPLIT_CASES:
            assert _run(hip_lib, "cuda", M, N, K, a_kc, b_kc, 4) < 2e-6, (a_kc, b_kc, M, N, K)
        assert _run(hip_lib, "cuda", 300, 132, 72, a_kc, b_kc, 4, epi=True) < 2e-6
    _splitk4(hip_lib, "cuda")
    _splitk4(hip_lib, "cuda", M=384, N=384, K=40000, ks=64)
    errs = _split_accuracy(hip_lib, "cuda", 1024, 384, 384)
    print("split-bf16 vs fp32-MFMA max error / (|A||B|):", errs)
    # the automatic plan takes the split path for the pair-level shapes (and only when operands allow)
    assert _run(hip_lib, "cuda", 65536, 128, 128, True, True, 0) < 2e-6


def _splitk4_rowsum(lib, dev, M=384, N=136, K=3000, ks=6, tile=4):
    """dW = dY^T X on the split-bf16 kernel with the bias gradient (column sums of dY) fused into the producers"""
    g = torch.Generator().manual_seed(8)
    dY = torch.randn(K, M, generator=g)
    X = torch.randn(K, N, generator=g)
    C = torch.zeros(M, N).to(dev)
    db0 = torch.randn(M, generator=g)
    db = db0.clone().to(dev)
    lib.gemm(dY.to(dev), X.to(dev), C, M, N, K, (1, M), (N, 1), N, ksplit=ks, tile=tile, a_rowsum=db)
    ref = dY.double().t() @ X.double()
    assert (C.cpu().double() - ref).abs().max() < 2e-6 * ref.abs().max()
    rs = db0.double() + dY.double().sum(0)
    assert (db.cpu().double() - rs).abs().max() < 1e-5 * rs.abs().max()


def test_gemm_split_rowsum_emu(emu_lib):
    _splitk4_rowsum(emu_lib, "cpu", M=300, N=72, K=200, ks=3)


@pytest.mark.gpu
def test_gemm_split_rowsum_gpu(hip_lib):
    _splitk4_rowsum(hip_lib, "cuda")
    _splitk4_rowsum(hip_lib, "cuda", M=384, N=384, K=65536, ks=64)


# ---- tile 6: the same split-bf16 kernel with a 128 x 128 block tile, two blocks per CU, two-stage LDS ring ----
@pytest.mark.parametrize("layout", SPLIT_LAYOUTS)
def test_gemm_split128_layouts_emu(emu_lib, layout):
    for (M, N, K) in SPLIT_CASES[:4]:
        assert _run(emu_lib, "cpu", M, N, K, layout[0], layout[1], 6) < 2e-6, (M, N, K)
    assert _run(emu_lib, "cpu", 152, 132, 72, layout[0], layout[1], 6, epi=True) < 2e-6


def test_gemm_split128_splitk_rowsum_emu(emu_lib):
    _splitk4(emu_lib, "cpu", tile=6)
    _splitk4_rowsum(emu_lib, "cpu", M=200, N=72, K=200, ks=3, tile=6)


@pytest.mark.gpu
def test_gemm_split128_gpu(hip_lib):
    for (a_kc, b_kc) in SPLIT_LAYOUTS:
        for (M, N, K) in SPLIT_CASES:
            assert _run(hip_lib, "cuda", M, N, K, a_kc, b_kc, 6) < 2e-6, (a_kc, b_kc, M, N, K)
        assert _run(hip_lib, "cuda", 300, 132, 72, a_kc, b_kc, 6, epi=True) < 2e-6
    _splitk4(hip_lib, "cuda", tile=6)
    _splitk4(hip_lib, "cuda", M=128, N=384, K=40000, ks=64, tile=6)
    _splitk4_rowsum(hip_lib, "cuda", tile=6)
    _splitk4_rowsum(hip_lib, "cuda", M=128, N=128, K=65536, ks=64, tile=6)


# ---- tile 10: 64 x 64 split-bf16 kernel (node-level / attention GEMMs): every layout through the same loader, the
# row-contiguous operands through the LDS transpose read ----
S64_CASES = [(64, 64, 32), (128, 128, 64), (100, 72, 40), (260, 136, 100), (132, 64, 36), (256, 384, 96)]


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
def test_gemm_s64_layouts_emu(emu_lib, a_kc, b_kc):
    for (M, N, K) in S64_CASES[:5]:
        assert _run(emu_lib, "cpu", M, N, K, a_kc, b_kc, 10) < 2e-6, (M, N, K)
    assert _run(emu_lib, "cpu", 100, 72, 40, a_kc, b_kc, 10, epi=True) < 2e-6


def _s64_plan(lib, dev):
    """the automatic plan sends aligned mid-size GEMMs to tile 10 and keeps unaligned ones on the fp32 tile"""
    from se3_diffusion_amd import hip
    import ctypes
    d = hip.FdGemmDesc()
    A = torch.zeros(3840 * 320, device=dev)
    d.A, d.B, d.C = A.data_ptr(), A.data_ptr(), A.data_ptr()
    d.M, d.N, d.K = 3840, 320, 320
    d.a_rs, d.a_cs, d.b_rs, d.b_cs, d.ldc = 320, 1, 1, 320, 320
    d.alpha = 1.0
    assert lib.cdll.fd_gemm_plan(ctypes.byref(d)) == 10
    d.K = 322                                   # K % 4 != 0: element-wise staging only exists on the fp32 tiles
    d.a_rs = 322
    assert lib.cdll.fd_gemm_plan(ctypes.byref(d)) == 2


def test_gemm_s64_plan_splitk_rowsum_emu(emu_lib):
    _s64_plan(emu_lib, "cpu")
    _splitk4(emu_lib, "cpu", tile=10)
    _splitk4_rowsum(emu_lib, "cpu", M=200, N=72, K=200, ks=3, tile=10)


@pytest.mark.gpu
def test_gemm_s64_gpu(hip_lib):
    for a_kc in (True, False):
        for b_kc in (True, False):
            for (M, N, K) in S64_CASES:
                assert _run(hip_lib, "cuda", M, N, K, a_kc, b_kc, 10) < 2e-6, (a_kc, b_kc, M, N, K)
            assert _run(hip_lib, "cuda", 300, 132, 72, a_kc, b_kc, 10, epi=True) < 2e-6
    _s64_plan(hip_lib, "cuda")
    _splitk4(hip_lib, "cuda", tile=10)
    _splitk4(hip_lib, "cuda", M=320, N=320, K=3840, ks=15, tile=10)
    _splitk4_rowsum(hip_lib, "cuda", tile=10)
    _splitk4_rowsum(hip_lib, "cuda", M=256, N=2688, K=3840, ks=13, tile=10)
    assert _run(hip_lib, "cuda", 3840, 320, 320, True, True, 0) < 2e-6
    assert _run(hip_lib, "cuda", 3840, 320, 960, True, False, 0) < 2e-6
