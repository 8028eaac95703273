"""Empty inputs: every family of entry points returns FD_OK for zero rows / zero backbones without launching and without
touching its outputs (a length-batched loader can hand a rank an empty shard: data/pdb_data_loader.py:300-352)."""
import torch

from se3_diffusion_amd import ops
from se3_diffusion_amd.ops import lib, mv


def _run(dev):
    L = lib()
    nan = lambda *s: torch.full(s, float("nan"), device=dev)
    z = lambda *s: torch.zeros(*s, device=dev)
    # GEMM family: M = 0 rows
    W, b, out = z(8, 4), z(8), nan(1, 8)
    ops.linear(mv(z(1, 4)), mv(W), b, mv(out), 0, 8, 4)
    assert torch.isnan(out).all()
    # LayerNorm / reductions
    y = nan(1, 128)
    ops.layernorm(mv(z(1, 128)), z(128), z(128), mv(y), 0, 128)
    assert torch.isnan(y).all()
    acc = nan(128)
    ops.bias_grad(mv(z(1, 128)), acc, 0, 128)
    assert torch.isnan(acc).all()
    # grouped pair-row weight gradients: rows = 0
    C = nan(384, 128)
    ops.pair_dw([dict(A=(z(1, 384), 0, 384), B=(z(1, 128), 0, 128), C=(C, 0, 128))], 0)
    assert torch.isnan(C).all()
    # fused edge transition: rows = 0 in every shape (2 = the column-split inference kernel, 4, 8), and the shapes' argument errors
    from se3_diffusion_amd import hip, options
    img = ops.edge_mlp_pack(z(384, 384), z(384, 384), z(128, 384))
    for shape in (0, 2, 4, 8):
        eo = nan(1, 128)
        with options.override(edge_shape=shape):
            ops.edge_mlp(z(1, 128), img, eo, 0, 1, p1=z(1, 384), q1=z(1, 384), bias2=z(384), pf=z(1, 128), qf=z(1, 128), gamma=z(128),
                         beta=z(128))
        assert torch.isnan(eo).all()
    d = hip.FdEdgeMlpDesc()
    for k, t in dict(x=z(1, 128), img=img, out=nan(1, 128), gmask1=z(1, 12).int(), gmask2=z(1, 12).int(), save1=nan(1, 384),
                     save2=nan(1, 384)).items():
        setattr(d, k, t.data_ptr())
    d.rows, d.nres, d.backward, d.eps, d.shape = 1, 1, 1, 1e-5, 2
    assert L.cdll.fd_edge_mlp(hip.ctypes.byref(d), None) != 0          # shape 2 is an inference forward
    assert "shape 2" in L.cdll.fd_last_error().decode()
    d.shape = 3
    assert L.cdll.fd_edge_mlp(hip.ctypes.byref(d), None) != 0 and "shape is 0" in L.cdll.fd_last_error().decode()
    # IPA attention / sequence attention: B = 0
    S, feats = nan(1, 8, 1, 1), nan(1, 2688)
    L.call("fd_ipa_attn_fwd", S, z(1, 40), z(1, 8, 24), z(1, 8, 24), None, z(8), z(1), feats, 0, 1)
    assert torch.isnan(S).all() and torch.isnan(feats).all()
    feats2, A2 = nan(1, 2688), nan(1, 8, 1, 1)
    L.call("fd_ipa_flash_fwd", z(1, 6816), z(1, 40), z(1, 8, 24), z(1, 8, 24), z(1, 8, 36), z(8), z(1), z(1, 4), z(1, 3), feats2, A2,
           0, 1, 0)
    assert torch.isnan(feats2).all() and torch.isnan(A2).all()
    dL, dzb, dqp, dkp, dhw = nan(1, 8, 1, 1), nan(1, 40), nan(1, 8, 24), nan(1, 8, 24), nan(8)
    L.call("fd_ipa_flash_bwd", z(1, 6816), z(1, 8, 1, 1), z(1, 40), z(1, 2688), z(1, 2688), z(1, 8, 36), z(1, 8), z(1, 8, 24),
           z(1, 8, 24), z(1, 8, 36), z(8), z(1, 3), dL, dzb, dqp, dkp, dhw, z(1, 8), 0, 1)
    assert all(torch.isnan(t).all() for t in (dL, dzb, dqp, dkp, dhw))
    o = nan(1, 320)
    L.call("fd_seq_attn_fwd", z(1, 960), None, o, None, 1.0, 0, 1)
    assert torch.isnan(o).all()
    dz = nan(1, 128)
    L.call("fd_ipa_dz_acc", z(1, 40), z(40, 128), dz, 0, 1)
    assert torch.isnan(dz).all()


def test_empty_emu(use_emu):
    _run("cpu")

# (host-side early returns only: the interpreter build runs the same entry-point code as the gfx950 library)
