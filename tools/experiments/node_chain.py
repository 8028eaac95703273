"""ctypes binding of tools/experiments/fd_node_chain.hip (built into libfd_experiments.so by build.py): GPU only."""
import ctypes
import os
from ctypes import Structure, c_float, c_int, c_long, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class FdNodeChainDesc(Structure):
    _fields_ = [
        ("x", c_void_p), ("img", c_void_p), ("out", c_void_p), ("bias", c_void_p * 3), ("save", c_void_p * 2),
        ("gate", c_void_p * 2), ("pre", c_void_p), ("ln_in", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
        ("rowscale", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("dgamma", c_void_p), ("dbeta", c_void_p),
        ("rows", c_long), ("width", c_int), ("nlayers", c_int), ("backward", c_int), ("eps", c_float), ("blocks", c_int),
    ]


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libfd_experiments.so")
        if not os.path.exists(path):
            import build
            build.build()
        _lib = ctypes.CDLL(path)
        _lib.fd_node_chain_pack.argtypes = [c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_void_p]
        _lib.fd_node_chain.argtypes = [c_void_p, c_void_p]
        _lib.fd_last_error.restype = ctypes.c_char_p
    return _lib


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def layer_bytes(width):
    return (width // 32) * (width // 64) * 12288


def node_chain_pack(weights, backward=False, out=None):
    """weights = [W_1, .., W_nl], each [W, W] contiguous ([out, in]); backward: the transposed chain W_nl^T .. W_1^T."""
    Wd = weights[0].shape[0]
    lb = layer_bytes(Wd)
    img = out if out is not None else torch.empty(len(weights) * lb, dtype=torch.uint8, device=weights[0].device)
    order = list(reversed(weights)) if backward else list(weights)
    for l, Wl in enumerate(order):
        assert Wl.is_contiguous() and tuple(Wl.shape) == (Wd, Wd)
        rs, cs = (1, Wd) if backward else (Wd, 1)
        rc = lib().fd_node_chain_pack(Wl.data_ptr(), rs, cs, Wd, int(backward or l > 0), img.data_ptr() + l * lb, _stream())
        assert rc == 0, lib().fd_last_error()
    return img


def node_chain(x, img, out, rows, width, nlayers, *, gamma, beta=None, bias=(), save=(), gate=(), pre=None, ln_in=None,
               rowscale=None, mean=None, rstd=None, dgamma=None, dbeta=None, backward=False, blocks=0):
    d = FdNodeChainDesc()
    for name, t in (("x", x), ("img", img), ("out", out), ("pre", pre), ("ln_in", ln_in), ("gamma", gamma), ("beta", beta),
                    ("rowscale", rowscale), ("mean", mean), ("rstd", rstd), ("dgamma", dgamma), ("dbeta", dbeta)):
        setattr(d, name, None if t is None else t.data_ptr())
    for field, seq in (("bias", bias), ("save", save), ("gate", gate)):
        for i, t in enumerate(seq):
            getattr(d, field)[i] = None if t is None else t.data_ptr()
    d.rows, d.width, d.nlayers, d.backward, d.eps, d.blocks = int(rows), int(width), int(nlayers), int(bool(backward)), 1e-5, int(blocks)
    rc = lib().fd_node_chain(ctypes.byref(d), _stream())
    assert rc == 0, lib().fd_last_error()
