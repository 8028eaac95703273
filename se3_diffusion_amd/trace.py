"""roctx ranges around the stages of the hot path (tracing row): with FD_ROCTX=1 every stage of
ScoreNetwork.forward / backward and of the sampler pushes a named range (libroctx64: roctxRangePushA / roctxRangePop) that
`rocprofv3 --marker-trace` shows on the timeline next to the kernels it launched.  Off by default: rng() then returns a shared
no-op context manager (no allocation, no ctypes call on the launch path)."""
from __future__ import annotations

import ctypes
import os
from contextlib import contextmanager

_LIB = None
_ON = os.environ.get("FD_ROCTX", "0") not in ("", "0")


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _Null()


def _lib():
    global _LIB, _ON
    if _LIB is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                _LIB = ctypes.CDLL(name)
                _LIB.roctxRangePushA.argtypes = [ctypes.c_char_p]
                break
            except OSError:
                _LIB = None
        if _LIB is None:
            _ON = False
    return _LIB


@contextmanager
def _range(name):
    lib = _lib()
    if lib is None:
        yield
        return
    lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        lib.roctxRangePop()


def rng(name):
    """`with rng("ipa_2.fwd"): ...` -- a roctx range when FD_ROCTX=1, nothing otherwise."""
    return _range(name) if _ON else _NULL


def enable(on=True):
    global _ON
    _ON = bool(on)
