"""ctypes binding of the C ABI declared in include/fd_hip.h.

The product path loads exactly one library: ``se3_diffusion_amd/lib/libfd_hip.so``
(hipcc, gfx950).  There is no CPU fallback: if the library is missing, or a
kernel is asked to run on a tensor that is not on an AMD GPU, we raise.

(The test suite can hand a *different* FdLib to individual calls -- the host
SIMT interpreter build under tests/emu -- to check kernel logic without a GPU.
That hook lives in tests/, is never selected here, and the product singleton
refuses any backend other than "gfx950".)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfd_hip.so")


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class FdError(RuntimeError):
    pass


class FdGemmDesc(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("a_rs", c_long), ("a_cs", c_long),
        ("b_rs", c_long), ("b_cs", c_long),
        ("ldc", c_long),
        ("batch", c_int), ("bdiv", c_int),
        ("a_so", c_long), ("a_si", c_long), ("b_so", c_long), ("b_si", c_long),
        ("c_so", c_long), ("c_si", c_long),
        ("alpha", c_float), ("beta", c_int),
        ("bias", c_void_p),
        ("pair_p", c_void_p), ("pair_q", c_void_p), ("ld_pair", c_long), ("nres", c_int),
        ("resid", c_void_p), ("ld_resid", c_long),
        ("gate", c_void_p), ("ld_gate", c_long),
        ("rowscale", c_void_p),
        ("relu", c_int), ("tile", c_int), ("ksplit", c_int), ("mtiles", c_int),
        ("a_rowsum", c_void_p),
        ("b_planes", c_void_p), ("b_plane_stride", c_long),
    ]


class FdHeadConst(Structure):
    _fields_ = [
        ("atoms", c_float * 15), ("Rd", c_float * 9), ("td", c_float * 3),
        ("coord_scale", c_float),
        ("exp_max_sigma", c_double), ("exp_min_sigma", c_double),
        ("min_b", c_float), ("max_b", c_float),
        ("L", c_int),
        ("score_norms", c_void_p), ("omega_grid", c_void_p), ("n_omega", c_int),
    ]


class FdLossDesc(Structure):
    _fields_ = [
        ("B", c_int), ("N", c_int),
        ("res_mask", c_void_p), ("fixed_mask", c_void_p), ("t", c_void_p),
        ("gt_trans_score", c_void_p), ("gt_rot_score", c_void_p),
        ("trans_score_scaling", c_void_p), ("rot_score_scaling", c_void_p),
        ("gt_rigids", c_void_p), ("gt_atom37", c_void_p),
        ("rot_score", c_void_p), ("trans_score", c_void_p), ("rigids", c_void_p), ("atom37", c_void_p),
        ("coordinate_scaling", c_float), ("trans_x0_threshold", c_float), ("trans_loss_weight", c_float),
        ("rot_loss_weight", c_float), ("rot_loss_t_threshold", c_float),
        ("bb_atom_loss_weight", c_float), ("bb_atom_loss_t_filter", c_float), ("aux_loss_weight", c_float),
        ("dist_mat_loss_weight", c_float), ("dist_mat_loss_t_filter", c_float),
        ("d_rot_score", c_void_p), ("d_trans_score", c_void_p), ("d_rigids", c_void_p), ("d_atom37", c_void_p),
        ("terms", c_void_p), ("loss", c_void_p), ("scratch", c_void_p),
        ("joint_rot_loss", c_int),
    ]


class FdEdgeMlpDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("img", c_void_p), ("p1", c_void_p), ("q1", c_void_p), ("bias2", c_void_p),
        ("save1", c_void_p), ("save2", c_void_p),
        ("pf", c_void_p), ("qf", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("rowscale", c_void_p),
        ("y", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("out", c_void_p),
        ("rows", c_long), ("nres", c_int), ("backward", c_int), ("eps", c_float), ("blocks", c_int),
        ("ld_pq", c_long), ("ld_pqf", c_long), ("zb_out", c_void_p), ("zb_bias", c_void_p),
        ("mask1", c_void_p), ("mask2", c_void_p), ("gmask1", c_void_p), ("gmask2", c_void_p),
        ("ln_y", c_void_p), ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("ln_gamma", c_void_p), ("ln_rowscale", c_void_p),
        ("dy_out", c_void_p), ("ln_dgamma", c_void_p), ("ln_dbeta", c_void_p), ("dzb", c_void_p), ("sched", c_void_p),
        ("shape", c_int),
    ]


EDGE_MLP_W8_MIN_ROWS = 65536
EDGE_MLP_PAIR_MAX_ROWS = 16384      # include/fd_hip.h FD_EDGE_MLP_PAIR_MAX_ROWS
EDGE_MLP_IMAGE_BYTES = 124 * 12288


class FdEdgeEmbedDesc(Structure):
    _fields_ = [
        ("seq_idx", c_void_p), ("sc_ca", c_void_p), ("idenom", c_void_p), ("dg_lower", c_void_p), ("dg_upper", c_void_p),
        ("img", c_void_p), ("p", c_void_p), ("q", c_void_p), ("bias2", c_void_p), ("bias3", c_void_p),
        ("gamma", c_void_p), ("beta", c_void_p), ("rowscale", c_void_p),
        ("h1", c_void_p), ("h2", c_void_p), ("h3", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("out", c_void_p),
        ("rows", c_long), ("nres", c_int), ("eps", c_float), ("blocks", c_int), ("zb_out", c_void_p), ("zb_bias", c_void_p),
        ("mask1", c_void_p), ("mask2", c_void_p), ("ld_pq", c_long),
    ]


EDGE_EMBED_IMAGE_BYTES = 24 * 12288


class FdEdgeEmbedBwdDesc(Structure):
    _fields_ = [
        ("dy", c_void_p), ("h3", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("gamma", c_void_p), ("rowscale", c_void_p),
        ("h2", c_void_p), ("h1", c_void_p), ("img", c_void_p), ("dh3", c_void_p), ("dh2", c_void_p), ("dh1", c_void_p),
        ("dgamma", c_void_p), ("dbeta", c_void_p), ("rows", c_long), ("blocks", c_int), ("gmask2", c_void_p), ("gmask1", c_void_p),
        ("sched", c_void_p),
    ]


EDGE_EMBED_BWD_IMAGE_BYTES = 16 * 12288

class FdLnGemmDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("ldx", c_long), ("gamma", c_void_p), ("beta", c_void_p), ("ln_rowscale", c_void_p),
        ("ln_out", c_void_p), ("ld_ln_out", c_long), ("W", c_void_p), ("ldw", c_long), ("bias", c_void_p),
        ("resid", c_void_p), ("ld_resid", c_long), ("out", c_void_p), ("ldo", c_long),
        ("M", c_int), ("N", c_int), ("K", c_int), ("relu", c_int), ("eps", c_float), ("ln_cols", c_int),
    ]


PAIR_DW_MAX_ITEMS = 8


class FdPairDwItem(Structure):
    _fields_ = [
        ("A", c_void_p), ("lda", c_long), ("A_add", c_void_p), ("ld_add", c_long), ("B", c_void_p), ("ldb", c_long),
        ("C", c_void_p), ("ldc", c_long), ("a_colsum", c_void_p), ("trans", c_int), ("a_bands", c_int), ("b_cols", c_int),
    ]


class FdPairDwDesc(Structure):
    _fields_ = [("item", FdPairDwItem * PAIR_DW_MAX_ITEMS), ("nitems", c_int), ("rows", c_long), ("blocks", c_int)]


GROUP_DW_MAX_ITEMS = 32


class FdGroupDwItem(Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("a_colsum", c_void_p),
                ("lda", c_int), ("ldb", c_int), ("ldc", c_int), ("n_out", c_int), ("k_in", c_int)]


class FdGroupDwDesc(Structure):
    _fields_ = [("item", FdGroupDwItem * GROUP_DW_MAX_ITEMS), ("nitems", c_int), ("rows", c_long), ("blocks", c_int)]


ABI_VERSION = 2          # FD_ABI_VERSION of include/fd_hip.h (bumped whenever a descriptor layout or a signature changes)


def _ptr(t, off=0):
    """Raw address of a tensor (plus an element offset)."""
    if t is None:
        return None
    if isinstance(t, tuple):
        t, off = t
    return t.data_ptr() + t.element_size() * int(off)


# Signature codes: p = device pointer (tensor | (tensor, elem_offset) | None), i = int, l = long,
# f = float, S = pointer to a ctypes struct, s = stream (supplied by the binding).
_SIGS = {
    "fd_gemm": "Ss",
    "fd_gemm_plan": "S",
    "fd_split_planes": "plps",
    "fd_gemm_set_exact_f32": "i",
    "fd_gemm_set_persistent_blocks": "i",
    "fd_edge_mlp_pack": "ppplps",
    "fd_edge_mlp": "Ss",
    "fd_edge_mlp_pack_zb": "pps",
    "fd_edge_mlp_pack_bwd": "ppplpps",
    "fd_edge_embed_pack": "pppps",
    "fd_edge_embed": "Ss",
    "fd_edge_embed_pack_zb": "pps",
    "fd_edge_embed_bwd_pack": "ppps",
    "fd_edge_embed_bwd": "Ss",
    "fd_ln_gemm": "Ss",
    "fd_pair_dw": "Ss",
    "fd_group_dw": "Ss",
    "fd_layernorm_fwd": "plpppplpplifs",
    "fd_layernorm_bwd": "plplpppppl" + "ipplis",
    "fd_colsum_acc": "pllips",
    "fd_pair_reduce_acc": "piiippls",
    "fd_axpby": "ppffls",
    "fd_rowscale": "plppllis",
    "fd_add2d": "plpllifs",
    "fd_node_feats": "ppppppiis",
    "fd_node_feats_ld": "ppppppliis",
    "fd_edge_feats": "pppppppppiis",
    "fd_ipa_points_fwd": "pppppppiliiiis",
    "fd_ipa_points_bwd": "pppppppliiiis",
    "fd_ipa_softmax_fwd": "ppppppiis",
    "fd_ipa_attn_fwd": "ppppppppiis",
    "fd_ipa_flash_fwd": "ppppppppppp" + "iiis",
    "fd_ipa_flash_fwd_split": "ppppppppppp" + "iiii" + "ps",
    "fd_ipa_flash_bwd": "pppppppppppp" + "pppppp" + "iis",
    "fd_ipa_flash_bwd_keys": "pppppppp" + "ppp" + "iiis",
    "fd_ipa_opt_bwd_dot": "ppppppp" + "ls",
    "fd_seq_attn_fwd": "ppppfiis",
    "fd_seq_attn_bwd": "pppppfiis",
    "fd_ipa_attn_bwd": "pppppppppppppiis",
    "fd_ipa_softmax_bwd": "ppppppppppiis",
    "fd_ipa_kpts_bwd": "pppppiis",
    "fd_ipa_dz_acc": "ppplis",
    "fd_ipa_opt_fwd": "ppppls",
    "fd_ipa_opt_bwd": "pppppls",
    "fd_ipa_opair_fwd": "pppiis",
    "fd_ipa_opair_bwd": "pppppiis",
    "fd_row_softmax_fwd": "pplii" + "s",
    "fd_row_softmax_bwd": "pplis",
    "fd_split_rigids": "pfpfppplis",
    "fd_bb_update_fwd": "plipppppppp" + "ls",
    "fd_bb_update_bwd": "pppppppppp" + "ls",
    "fd_heads_fwd": "pppp" + "pl" + "ppp" + "pi" + "S" + "ppppppp" + "iis",
    "fd_heads_bwd": "ppppp" + "ppp" + "pi" + "S" + "ppppp" + "ppp" + "iis",
    "fd_backbone_atoms": "ppSppls",
    "fd_igso3_tables": "ppiiippps",
    "fd_sample_ref": "pppppidpls",
    "fd_forward_marginal": "ppppppipdddipppp" + "ls",
    "fd_forward_marginal_batch": "ppppppippdipppp" + "iis",
    "fd_se3_reverse_step": "ppppppiiddpdddiiips",
    "fd_se3_reverse_step_f32": "ppppppiiddpdddiiips",
    "fd_se3_reverse_step_net": "ppppppiiddpdddiiips",
    "fd_sample_advance": "ppppilpipps",
    "fd_dsm_loss": "Ss",
    "fd_adam_step": "pppplffffffs",
}
# exact argument lists, kept next to the header for the symbol-export test
_CT = {"p": c_void_p, "i": c_int, "l": c_long, "f": c_float, "d": c_double, "S": c_void_p, "s": c_void_p}


class FdLib:
    """One loaded build of the C ABI (product: gfx950)."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FdError(
                f"HIP extension not found at {path}. Build it with "
                f"`python -m se3_diffusion_amd.build` (hipcc, gfx950). There is no CPU fallback.")
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (res, args) in {"fd_last_error": (c_char_p, []), "fd_abi_version": (c_int, []), "fd_launch_count": (c_long, []),
                                  "fd_backend": (c_char_p, []), "fd_build_flags": (c_char_p, [])}.items():
            fn = getattr(self.cdll, name)
            fn.restype, fn.argtypes = res, args
        for name, sig in _SIGS.items():
            fn = getattr(self.cdll, name)  # AttributeError if the symbol is missing
            fn.restype = c_int
            fn.argtypes = [_CT[c] for c in sig]
        if self.cdll.fd_abi_version() != ABI_VERSION:
            # descriptors are passed by layout: a library built from another header revision would read them misaligned
            raise FdError(f"{path} implements C-ABI version {self.cdll.fd_abi_version()}, this binding is version {ABI_VERSION}: "
                          f"rebuild it (python -m se3_diffusion_amd.build)")
        self.backend = self.cdll.fd_backend().decode()
        self.is_device = self.backend == "gfx950"
        # mirror of the library's FD_GEMM_EXACT_F32 switch (the fused split-bf16 kernels consult it on the host side)
        self.exact_f32 = os.environ.get("FD_GEMM_EXACT_F32", "0") not in ("", "0")

    def set_exact_f32(self, exact: bool) -> bool:
        """exact=True: every GEMM on the fp32-MFMA kernels (bitwise fmaf chains) and the fused split-bf16 kernels off
        (fd_gemm_set_exact_f32).  Returns the previous setting."""
        was = self.exact_f32
        self.cdll.fd_gemm_set_exact_f32(1 if exact else 0)
        self.exact_f32 = bool(exact)
        return was

    # -- helpers ---------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise FdError(f"{what} failed ({rc}): {self.cdll.fd_last_error().decode()}")

    def _stream(self, tensors):
        if not self.is_device:
            for t in tensors:
                if t.is_cuda:
                    raise FdError("emulator build called with a GPU tensor")
            return None
        for t in tensors:
            if not t.is_cuda:
                raise FdError("HIP kernel called with a CPU tensor; the hot path has no CPU fallback")
        # (torch.cuda.current_stream() costs ~8 us of Python per call -- 5 ms of host time per training step; the raw
        # handle of the same current stream is a C call)
        if _RAW_STREAM is not None:
            return _RAW_STREAM(tensors[0].device.index if tensors else torch.cuda.current_device())
        return torch.cuda.current_stream().cuda_stream

    def call(self, name: str, *args):
        """Invoke an entry point; tensors become raw pointers, the stream is appended."""
        sig = _SIGS[name]
        if len(args) != len(sig) - 1:
            raise FdError(f"{name}: expected {len(sig) - 1} arguments, got {len(args)}")
        conv, tens = [], []
        for code, a in zip(sig, args):
            if code == "p":
                t = a[0] if isinstance(a, tuple) else a
                if t is not None:
                    if not t.is_contiguous() and t.dim() > 0 and not isinstance(a, tuple):
                        # strided views are addressed explicitly by the caller via ld/offset arguments
                        pass
                    tens.append(t)
                conv.append(_ptr(a))
            elif code == "S":
                conv.append(ctypes.addressof(a))
            elif code in "fd":
                conv.append(float(a))
            else:
                conv.append(int(a))
        conv.append(self._stream(tens))
        self._check(getattr(self.cdll, name)(*conv), name)

    # -- dense -----------------------------------------------------------
    def gemm(self, A, B, C, M, N, K, a_str, b_str, ldc, *, a_off=0, b_off=0, c_off=0,
             batch=1, bdiv=1, a_bs=(0, 0), b_bs=(0, 0), c_bs=(0, 0), alpha=1.0, beta=False,
             bias=None, pair=None, resid=None, ld_resid=0, gate=None, ld_gate=0,
             rowscale=None, relu=False, tile=0, ksplit=1, mtiles=0, a_rowsum=None, b_planes=None):
        """b_planes = (address of plane 0's element of B[b_off], elements between planes) when B lives in a buffer that
        fd_split_planes has split (ops.weight_planes): fd_gemm may then pick the pre-split tiles 12-14."""
        d = FdGemmDesc()
        d.A, d.B, d.C = _ptr(A, a_off), _ptr(B, b_off), _ptr(C, c_off)
        d.M, d.N, d.K = int(M), int(N), int(K)
        d.a_rs, d.a_cs = a_str
        d.b_rs, d.b_cs = b_str
        d.ldc = ldc
        d.batch, d.bdiv = batch, bdiv
        d.a_so, d.a_si = a_bs
        d.b_so, d.b_si = b_bs
        d.c_so, d.c_si = c_bs
        d.alpha, d.beta = float(alpha), int(bool(beta))
        d.bias = _ptr(bias)
        tens = [A, B, C]
        if pair is not None:
            P, Q, ld, nres = pair
            d.pair_p, d.pair_q, d.ld_pair, d.nres = _ptr(P), _ptr(Q), ld, nres
            tens += [P[0] if isinstance(P, tuple) else P, Q[0] if isinstance(Q, tuple) else Q]
        d.resid, d.ld_resid = _ptr(resid), ld_resid
        d.gate, d.ld_gate = _ptr(gate), ld_gate
        d.rowscale = _ptr(rowscale)
        d.relu, d.tile, d.ksplit, d.mtiles = int(bool(relu)), int(tile), int(ksplit), int(mtiles)
        d.a_rowsum = _ptr(a_rowsum)
        if b_planes is not None:
            d.b_planes, d.b_plane_stride = int(b_planes[0]), int(b_planes[1])
        for x in (bias, resid, gate, rowscale):
            if x is not None:
                tens.append(x[0] if isinstance(x, tuple) else x)
        stream = self._stream(tens)
        prof = self.gemm_profile
        if prof is not None and self.is_device:
            d.tile = self.cdll.fd_gemm_plan(ctypes.byref(d))
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self._check(self.cdll.fd_gemm(ctypes.byref(d), stream), "fd_gemm")
            e1.record()
            prof.append((d.tile, d.a_cs == 1, d.b_rs == 1, 2.0 * d.M * d.N * d.K * max(1, batch), e0, e1,
                         (d.M, d.N, d.K, max(1, batch), int(bool(d.gate)), d.beta, int(bool(d.pair_p)), d.ksplit)))
            return
        self._check(self.cdll.fd_gemm(ctypes.byref(d), stream), "fd_gemm")

    gemm_profile = None  # set to a list to record (tile, a_kc, b_kc, flops, ev0, ev1) per fd_gemm launch


_PRODUCT: FdLib | None = None
_TEST_OVERRIDE: FdLib | None = None  # set only by tests (never by product code)


def get_lib() -> FdLib:
    """The library every op dispatches to."""
    global _PRODUCT
    if _TEST_OVERRIDE is not None:
        return _TEST_OVERRIDE
    if _PRODUCT is None:
        lib = FdLib(LIB_PATH)
        if lib.backend != "gfx950":
            raise FdError(f"{LIB_PATH} reports backend {lib.backend!r}; the product path only runs gfx950 code")
        _PRODUCT = lib
    return _PRODUCT


def exported_symbols():
    return sorted(list(_SIGS) + ["fd_last_error", "fd_abi_version", "fd_backend", "fd_build_flags", "fd_launch_count"])
