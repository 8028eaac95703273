"""SE(3) diffuser parity against tests/golden/diffuser.npz (outputs of the UNMODIFIED reference with
its own numpy RNG stream, written by oracle/make_golden.py).

Frames are compared as rotation matrices / quaternions up to sign.  Tolerances: fp64 arithmetic stored
as fp32 frames -> 2e-6 absolute on unit quaternions / rotation matrices, 2e-5 A on translations;
scores 1e-9 relative (float64 series) except where the reference evaluates in float32."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd.data import se3_diffuser, utils as du  # noqa: E402
from se3_diffusion_amd.openfold.utils import rigid_utils as ru  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "diffuser.npz"))
CACHE = os.environ.get("FD_TEST_IGSO3_CACHE", "/tmp/fd_test_igso3_cache")


def conf(cache=CACHE):
    ns = SimpleNamespace
    return ns(diffuse_trans=True, diffuse_rot=True, r3=ns(min_b=0.1, max_b=20.0, coordinate_scaling=0.1),
              so3=ns(num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5, schedule="logarithmic",
                     cache_dir=cache, use_cached_score=False))


@pytest.fixture(scope="module")
def diff():
    return se3_diffuser.SE3Diffuser(conf())


def rotmats(t7):
    return du.quat_wxyz_to_matrix(np.asarray(t7[..., :4], dtype=np.float64))


def test_schedules_and_tables(diff):
    so3, r3 = diff._so3_diffuser, diff._r3_diffuser
    ts = G["ts"]
    assert np.allclose([so3.sigma(t) for t in ts], G["sigma"], rtol=1e-14)
    assert np.array_equal([so3.t_to_idx(t) for t in ts], G["t_to_idx"])
    assert np.allclose([so3.diffusion_coef(t) for t in ts], G["g_rot"], rtol=1e-13)
    assert np.allclose([r3.b_t(t) for t in ts], G["b_t"]) and np.allclose([r3.marginal_b_t(t) for t in ts], G["marginal_b_t"])
    assert np.allclose([r3.score_scaling(t) for t in ts], G["trans_score_scaling"], rtol=1e-13)
    rows, cols = G["tab_rows"], G["tab_cols"]
    assert np.allclose(so3._pdf[np.ix_(rows, cols)], G["pdf_sample"], rtol=1e-9, atol=1e-12)
    assert np.allclose(so3._cdf[np.ix_(rows, cols)], G["cdf_sample"], rtol=1e-9, atol=1e-12)
    assert np.allclose(so3._score_norms[np.ix_(rows, cols)], G["score_norms_sample"], rtol=1e-8, atol=2e-7)
    assert np.allclose(so3._cdf[499], G["cdf_row_499"], rtol=1e-9, atol=1e-12)
    assert np.allclose(so3._score_scaling[[0, 9, 499, 999]], G["score_scaling_table_sample"], rtol=1e-8)
    assert np.allclose([so3.score_scaling(t) for t in ts], G["rot_score_scaling"], rtol=1e-8)
    with pytest.raises(ValueError):
        so3.sigma(np.array(1.5))
    with pytest.raises(ValueError):
        r3.b_t(np.array(-0.1))


def test_torch_score_host_differentiable(diff):
    """SO3Diffuser.torch_score on host tensors that require a gradient (reference data/so3_diffuser.py:274-305 is differentiable
    on the CPU): value == the numpy path, gradient == central differences of the numpy path (float64)"""
    so3 = diff._so3_diffuser
    g = torch.Generator().manual_seed(5)
    v = (torch.randn(2, 6, 3, generator=g, dtype=torch.float64) * 0.6).requires_grad_(True)
    t = torch.tensor([0.3, 0.8])
    sc = so3.torch_score(v, t)
    ref = so3.torch_score(v.detach(), t)
    assert sc.dtype == torch.float64 and float((sc.detach() - ref).abs().max()) < 1e-10 * float(ref.abs().max())
    w = torch.randn(2, 6, 3, generator=g, dtype=torch.float64)
    (sc * w).sum().backward()
    h = 1e-6
    for (b, n, k) in ((0, 0, 0), (1, 3, 2), (0, 5, 1)):
        vp, vm = v.detach().clone(), v.detach().clone()
        vp[b, n, k] += h
        vm[b, n, k] -= h
        fd_ = float(((so3.torch_score(vp, t) - so3.torch_score(vm, t)) * w).sum()) / (2 * h)
        assert abs(float(v.grad[b, n, k]) - fd_) < 1e-5 * (abs(fd_) + 1.0), (b, n, k, float(v.grad[b, n, k]), fd_)


def test_torch_score_host(diff):
    so3 = diff._so3_diffuser
    sc = so3.torch_score(torch.tensor(G["ts_vec"]), torch.tensor(G["ts"], dtype=torch.float32)).numpy()
    ref = G["ts_score"]
    # the reference evaluates sin/cos((l+1/2) w) in float32: compare away from w -> 0 where that noise explodes
    # ... and where the density is not negligible: for omega >> sigma the true f is ~0 and the reference's
    # f'/(f + 1e-4) is float32 round-off of O(1) (a float64 evaluation gives ~0) -- see DESIGN.md "numerics".
    from se3_diffusion_amd.data.so3_diffuser import _series
    om = np.linalg.norm(G["ts_vec"], axis=-1)
    sg = so3.discrete_sigma[so3.t_to_idx(G["ts"].astype(np.float32).astype(np.float64))][:, None]
    f, _ = _series(om + 1e-6, sg)
    ok = (om > 0.05) & (f > 1e-2)
    assert ok.sum() >= 20
    assert (np.abs(sc - ref)[ok] < 5e-4 * np.abs(ref)[ok].max(-1, keepdims=True) + 1e-6).all()
    # calc_rot_score through Rotation objects (quaternion algebra in fp32)
    out = diff.calc_rot_score(ru.Rotation(quats=torch.tensor(G["crs_qt"]), normalize_quats=False),
                              ru.Rotation(quats=torch.tensor(G["crs_q0"]), normalize_quats=False),
                              torch.tensor(G["ts"], dtype=torch.float32)).numpy()
    # rows t >= 0.2 only: at t = 0.01 / 0.05 random relative rotations sit in the negligible-density regime above
    assert np.abs(out - G["crs_out"])[2:].max() < 5e-4 * np.abs(G["crs_out"])[2:].max()
    ts_out = diff.calc_trans_score(torch.tensor(G["cts_xt"]), torch.tensor(G["cts_x0"]),
                                   torch.tensor(G["ts"], dtype=torch.float32)[:, None, None], use_torch=True).numpy()
    assert np.allclose(ts_out, G["cts_out"], rtol=1e-5, atol=1e-6)


def _check_reverse(out7, tag):
    assert np.abs(rotmats(out7) - G[f"rev_{tag}_out_rotmats"]).max() < 3e-6
    assert np.abs(np.asarray(out7[..., 4:]) - G[f"rev_{tag}_out_trans"]).max() < 3e-5


def test_reverse_host_numpy_stream(diff):
    """Same numpy seed as the reference -> same frames (rotation noise drawn first, then translation)."""
    for tag in ("a", "b"):
        np.random.seed(123)
        out = diff.reverse(rigid_t=ru.Rigid.from_tensor_7(torch.tensor(G["rev_rigids"])), rot_score=G["rev_rot_score"],
                           trans_score=G["rev_trans_score"], t=float(G[f"rev_{tag}_t"]), dt=1 / 100,
                           diffuse_mask=G["rev_dmask"], center=True, noise_scale=float(G[f"rev_{tag}_ns"]))
        _check_reverse(out.to_tensor_7().numpy(), tag)
    with pytest.raises(ValueError):
        diff.reverse(rigid_t=ru.Rigid.from_tensor_7(torch.tensor(G["rev_rigids"])), rot_score=G["rev_rot_score"],
                     trans_score=G["rev_trans_score"], t=np.array([0.5, 0.5]), dt=0.01)


def _reverse_kernel(diff, dev):
    for tag in ("a", "b"):
        out = diff.reverse_device(torch.tensor(G["rev_rigids"]).to(dev), G["rev_rot_score"], G["rev_trans_score"],
                                  float(G[f"rev_{tag}_t"]), 1 / 100, diffuse_mask=G["rev_dmask"], center=True,
                                  noise_scale=float(G[f"rev_{tag}_ns"]), noise=(G[f"rev_{tag}_zrot"], G[f"rev_{tag}_ztrans"]))
        _check_reverse(out.cpu().numpy(), tag)


def test_reverse_kernel_emu(diff, use_emu):
    _reverse_kernel(diff, "cpu")


def _check_sample_ref(t7):
    ref = G["sr_out_t7"]
    assert np.abs(rotmats(t7) - rotmats(ref)).max() < 3e-6
    assert np.abs(np.asarray(t7[..., 4:]) - ref[..., 4:]).max() < 3e-5


def test_sample_ref_host(diff):
    np.random.seed(321)
    _check_sample_ref(diff.sample_ref(n_samples=11, as_tensor_7=True)["rigids_t"].numpy())
    with pytest.raises(ValueError):
        diff.sample_ref(n_samples=4, diffuse_mask=np.ones(4))


def test_sample_ref_kernel_emu(diff, use_emu):
    t7 = diff.sample_ref_device(11, "cpu", noise=(G["sr_randn"], G["sr_rand"], G["sr_normal"]))
    _check_sample_ref(t7.numpy())


def _check_fm(out, f32_scores=False):
    t7 = out["rigids_t"].cpu().numpy() if torch.is_tensor(out["rigids_t"]) else out["rigids_t"]
    assert np.abs(rotmats(t7) - rotmats(G["fm_rigids_t"])).max() < 3e-6
    assert np.abs(t7[..., 4:] - G["fm_rigids_t"][..., 4:]).max() < 3e-5
    # (the batch generator returns the scores as float32 training-batch entries)
    rt, at = (1e-6, 1e-6) if f32_scores else (1e-9, 1e-10)
    assert np.allclose(out["trans_score"], G["fm_trans_score"], rtol=rt, atol=at)
    assert np.allclose(out["rot_score"], G["fm_rot_score"], rtol=max(rt, 1e-7), atol=max(at, 1e-9))
    rs = 1e-6 if f32_scores else 1e-8
    assert np.allclose(out["trans_score_scaling"], G["fm_trans_score_scaling"], rtol=max(rs, 1e-7))
    assert np.allclose(out["rot_score_scaling"], G["fm_rot_score_scaling"], rtol=rs)


def test_forward_marginal_host(diff):
    np.random.seed(55)
    _check_fm(diff.forward_marginal(ru.Rigid.from_tensor_7(torch.tensor(G["fm_rigids0"])), float(G["fm_t"])))


def _fm_kernel(diff, dev):
    from se3_diffusion_amd import hip
    so3, r3 = diff._so3_diffuser, diff._r3_diffuser
    t = float(G["fm_t"])
    n = 10
    cdf, omega = so3.device_tables(dev)
    idx = int(so3.t_to_idx(t))
    r0 = torch.tensor(G["fm_rigids0"]).to(dev)
    rt = torch.empty_like(r0)
    rs = torch.empty((n, 3), dtype=torch.float64, device=dev)
    ts = torch.empty((n, 3), dtype=torch.float64, device=dev)
    f64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    hip.get_lib().call("fd_forward_marginal", r0, f64(G["fm_randn"]), f64(G["fm_rand"]), f64(G["fm_normal"]),
                       (cdf, idx * cdf.shape[1]), omega, omega.numel(), None, float(so3.discrete_sigma[idx]),
                       float(r3.marginal_b_t(t)), 0.1, 1000, None, rt, rs, ts, n)
    _check_fm(dict(rigids_t=rt, trans_score=ts.cpu().numpy(), rot_score=rs.cpu().numpy(),
                   trans_score_scaling=r3.score_scaling(t), rot_score_scaling=so3.score_scaling(t)))


def test_forward_marginal_kernel_emu(diff, use_emu):
    _fm_kernel(diff, "cpu")


def _fm_batch(diff, dev):
    """forward_marginal_batch (device training-batch generation, per-example t) == the golden single-example result
    for the example that carries the golden noise, and per-example calls for the others"""
    n = 10
    r0 = torch.tensor(G["fm_rigids0"])
    rs = np.random.RandomState(4)
    B = 3
    z_axis = rs.standard_normal((B, n, 3)); u = rs.uniform(size=(B, n)); z_trans = rs.standard_normal((B, n, 3))
    z_axis[1], u[1], z_trans[1] = G["fm_randn"], G["fm_rand"], G["fm_normal"]
    t = np.array([0.13, float(G["fm_t"]), 0.77])
    out = diff.forward_marginal_batch(r0[None].repeat(B, 1, 1).to(dev), t, noise=(z_axis, u, z_trans))
    so3, r3 = diff._so3_diffuser, diff._r3_diffuser
    _check_fm(dict(rigids_t=out["rigids_t"][1], trans_score=out["trans_score"][1].double().cpu().numpy(),
                   rot_score=out["rot_score"][1].double().cpu().numpy(),
                   trans_score_scaling=float(out["trans_score_scaling"][1]), rot_score_scaling=float(out["rot_score_scaling"][1])),
              f32_scores=True)
    for b in (0, 2):
        assert abs(float(out["rot_score_scaling"][b]) - so3.score_scaling(t[b])) < 1e-5 * so3.score_scaling(t[b])
        assert abs(float(out["trans_score_scaling"][b]) - r3.score_scaling(t[b])) < 1e-5 * r3.score_scaling(t[b])
        assert torch.isfinite(out["rigids_t"][b]).all() and not torch.equal(out["rigids_t"][b], out["rigids_t"][1])


def test_forward_marginal_batch_emu(diff, use_emu):
    _fm_batch(diff, "cpu")


GL = np.load(os.path.join(ROOT, "tests", "golden", "diffuser_large.npz"))


def _large(diff, dev):
    """The diffuser kernels at shipped sizes against the unmodified reference (fixture diffuser_large.npz, written by
    oracle/make_golden_diffuser_large.py: B=3 x N=300 reverse steps with fixed residues at dt = 1/500, a 700-residue prior draw,
    an N=300 forward marginal); the reference's numpy draws are re-created from the stored seeds in its call order (rotation
    first, then translation: se3_diffuser.py:213-262)."""
    from se3_diffusion_amd import hip
    from oracle.make_golden_diffuser_large import inputs
    x = inputs()
    B, N = x["B"], x["N"]
    for tag in ("a", "b"):
        np.random.seed(int(GL[f"rev_{tag}_seed"]))
        z_rot, z_trans = np.random.normal(size=(B, N, 3)), np.random.normal(size=(B, N, 3))
        out = diff.reverse_device(x["rig"].to(dev), x["rot_score"], x["trans_score"], float(GL[f"rev_{tag}_t"]), 1 / 500,
                                  diffuse_mask=x["dmask"], center=True, noise_scale=float(GL[f"rev_{tag}_ns"]),
                                  noise=(z_rot, z_trans)).cpu().numpy()
        ref = GL[f"rev_{tag}_out_t7"]
        assert np.abs(rotmats(out) - rotmats(ref)).max() < 3e-6 and np.abs(out[..., 4:] - ref[..., 4:]).max() < 5e-5, tag
    n = int(GL["sr_n"])
    np.random.seed(int(GL["sr_seed"]))
    draws = (np.random.randn(n, 3), np.random.rand(n), np.random.normal(size=(n, 3)))
    t7 = diff.sample_ref_device(n, dev, noise=draws).cpu().numpy()
    assert np.abs(rotmats(t7) - rotmats(GL["sr_out_t7"])).max() < 3e-6 and np.abs(t7[..., 4:] - GL["sr_out_t7"][..., 4:]).max() < 5e-5
    so3, r3 = diff._so3_diffuser, diff._r3_diffuser
    t = float(GL["fm_t"])
    np.random.seed(int(GL["fm_seed"]))
    fz = (np.random.randn(N, 3), np.random.rand(N), np.random.normal(size=(N, 3)))
    cdf, omega = so3.device_tables(dev)
    idx = int(so3.t_to_idx(t))
    r0 = x["rig0"].to(dev)
    rt = torch.empty_like(r0)
    rs = torch.empty((N, 3), dtype=torch.float64, device=dev)
    tsc = torch.empty((N, 3), dtype=torch.float64, device=dev)
    f64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    hip.get_lib().call("fd_forward_marginal", r0, f64(fz[0]), f64(fz[1]), f64(fz[2]), (cdf, idx * cdf.shape[1]), omega,
                       omega.numel(), None, float(so3.discrete_sigma[idx]), float(r3.marginal_b_t(t)), 0.1, 1000, None, rt, rs, tsc, N)
    got = rt.cpu().numpy()
    assert np.abs(rotmats(got) - rotmats(GL["fm_rigids_t"])).max() < 3e-6
    assert np.abs(got[..., 4:] - GL["fm_rigids_t"][..., 4:]).max() < 5e-5
    assert np.allclose(tsc.cpu().numpy(), GL["fm_trans_score"], rtol=1e-6, atol=1e-6)
    assert np.allclose(rs.cpu().numpy(), GL["fm_rot_score"], rtol=1e-6, atol=1e-8)


def test_diffuser_kernels_large_emu(diff, use_emu):
    _large(diff, "cpu")


def test_igso3_tables_kernel_emu(diff, use_emu):
    """fd_igso3_tables on a sub-grid vs the cached full tables."""
    from se3_diffusion_amd import hip
    so3 = diff._so3_diffuser
    rows = np.array([0, 9, 499, 999])
    sg = torch.tensor(so3.discrete_sigma[rows], dtype=torch.float64)
    om = torch.tensor(so3.discrete_omega[:64], dtype=torch.float64)
    pdf = torch.empty((4, 64), dtype=torch.float64); cdf = torch.empty_like(pdf); sn = torch.empty_like(pdf)
    hip.get_lib().call("fd_igso3_tables", sg, om, 4, 64, 1000, pdf, cdf, sn)
    assert np.allclose(pdf.numpy(), so3._pdf[rows][:, :64], rtol=1e-9, atol=1e-13)
    assert np.allclose(sn.numpy(), so3._score_norms[rows][:, :64], rtol=1e-8, atol=2e-7)
    assert np.allclose(cdf.numpy() * 64 / 1000, so3._cdf[rows][:, :64], rtol=1e-9, atol=1e-13)


@pytest.mark.gpu
def test_diffuser_kernels_gpu(hip_lib, tmp_path):
    d = se3_diffuser.SE3Diffuser(conf(str(tmp_path)))      # builds the tables with fd_igso3_tables on the GPU
    test_schedules_and_tables(d)
    _reverse_kernel(d, "cuda")
    _check_sample_ref(d.sample_ref_device(11, "cuda", noise=(G["sr_randn"], G["sr_rand"], G["sr_normal"])).cpu().numpy())
    _fm_kernel(d, "cuda")
    _fm_batch(d, "cuda")
    _large(d, "cuda")
    np.random.seed(55)
    _check_fm(d.forward_marginal(ru.Rigid.from_tensor_7(torch.tensor(G["fm_rigids0"]).cuda()), float(G["fm_t"])))
    np.random.seed(123)
    out = d.reverse(rigid_t=ru.Rigid.from_tensor_7(torch.tensor(G["rev_rigids"]).cuda()), rot_score=G["rev_rot_score"],
                    trans_score=G["rev_trans_score"], t=float(G["rev_a_t"]), dt=1 / 100, diffuse_mask=G["rev_dmask"],
                    center=True, noise_scale=float(G["rev_a_ns"]))
    _check_reverse(out.to_tensor_7().cpu().numpy(), "a")
    # differentiable calc_rot_score on the GPU vs the float64 host series
    qt, q0 = torch.tensor(G["crs_qt"]).cuda(), torch.tensor(G["crs_q0"]).cuda().requires_grad_(True)
    tt = torch.tensor(G["ts"], dtype=torch.float32).cuda()
    sc = d.calc_rot_score(ru.Rotation(quats=qt, normalize_quats=False), ru.Rotation(quats=q0, normalize_quats=False), tt)
    assert np.abs(sc.detach().cpu().numpy() - G["crs_out"])[2:].max() < 5e-4 * np.abs(G["crs_out"])[2:].max()
    # ... and its gradient w.r.t. q_0 (score_ops._RotScoreFn.backward = fd_heads_bwd) against central differences of the same
    # entry point: L = sum(w * score), h = 1e-3 on each of the 4 components of every residue's quaternion (fp32 input, fp64
    # output: truncation O(h^2) ~ 1e-6, round-off ~ 1e-7 / h = 1e-4 of the score's size); bound 2e-3 of the gradient's maximum
    w = torch.tensor(np.random.RandomState(7).randn(*sc.shape), dtype=torch.float64).cuda()
    (sc * w).sum().backward()
    assert torch.isfinite(q0.grad).all()
    g = q0.grad.detach().double().cpu().numpy()

    def loss_at(q):
        with torch.no_grad():
            o = d.calc_rot_score(ru.Rotation(quats=qt, normalize_quats=False), ru.Rotation(quats=q, normalize_quats=False), tt)
        return float((o * w).sum())
    h = 1e-3
    fd_g = np.zeros_like(g)
    base = q0.detach()
    for idx in np.ndindex(*g.shape):
        qp, qm = base.clone(), base.clone()
        qp[idx] += h
        qm[idx] -= h
        fd_g[idx] = (loss_at(qp) - loss_at(qm)) / (float(qp[idx]) - float(qm[idx]))
    err = np.abs(g - fd_g).max() / np.abs(fd_g).max()
    print(f"[parity] calc_rot_score backward vs central differences: max err {err:.2e} of max |grad| {np.abs(fd_g).max():.3e}")
    assert err < 2e-3, (err, g.reshape(-1)[:8], fd_g.reshape(-1)[:8])
