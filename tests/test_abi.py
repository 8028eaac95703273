"""C-ABI surface: include/fd_hip.h <-> ctypes binding <-> the built gfx950 library.
No compute calls (works without a GPU)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import hip  # noqa: E402


def header_symbols():
    src = open(os.path.join(ROOT, "include", "fd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", src)))


def test_binding_covers_header():
    assert header_symbols() == hip.exported_symbols()


def test_binding_constants_match_header():
    """the numeric constants hip.py mirrors (image sizes, item limits, the shape rule's row bounds) are the header's"""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fd_hip.h")).read()
    defs = {m.group(1): m.group(2) for m in re.finditer(r"#define (FD_[A-Z0-9_]+)\s+\(?([0-9 *L]+?)\)?\s*(?:/\*.*)?$", hdr, re.M)}
    val = lambda e: eval(e.replace("L", ""))           # "124 * 12288", "65536L"
    pairs = {"FD_EDGE_MLP_IMAGE_BYTES": hip.EDGE_MLP_IMAGE_BYTES, "FD_EDGE_MLP_W8_MIN_ROWS": hip.EDGE_MLP_W8_MIN_ROWS,
             "FD_EDGE_MLP_PAIR_MAX_ROWS": hip.EDGE_MLP_PAIR_MAX_ROWS, "FD_EDGE_EMBED_IMAGE_BYTES": hip.EDGE_EMBED_IMAGE_BYTES,
             "FD_EDGE_EMBED_BWD_IMAGE_BYTES": hip.EDGE_EMBED_BWD_IMAGE_BYTES, "FD_PAIR_DW_MAX_ITEMS": hip.PAIR_DW_MAX_ITEMS,
             "FD_GROUP_DW_MAX_ITEMS": hip.GROUP_DW_MAX_ITEMS}
    for name, py in pairs.items():
        assert name in defs, name
        assert val(defs[name]) == py, (name, defs[name], py)


def test_product_library_exports_every_symbol():
    if not os.path.exists(hip.LIB_PATH):
        from se3_diffusion_amd import build
        build.build(verbose=False)
    lib = hip.FdLib(hip.LIB_PATH)            # binds every symbol; AttributeError if one is missing
    assert lib.backend == "gfx950"
    assert lib.cdll.fd_abi_version() == hip.ABI_VERSION == 2
    # built without the probe switch: none of the timing / ablation hooks of the kernel sources (csrc/fd_probe.h) is in it
    assert lib.cdll.fd_build_flags() == b""
    for s in header_symbols():
        assert hasattr(lib.cdll, s)


def test_no_cpu_fallback(emu_lib, tmp_path):
    import torch
    # a CPU tensor handed to the gfx950 build is an error, not a silent fallback
    lib = hip.FdLib(hip.LIB_PATH)
    with pytest.raises(hip.FdError):
        lib.call("fd_axpby", torch.zeros(4), torch.zeros(4), 1.0, 1.0, 4)
    # a missing library is an error
    with pytest.raises(hip.FdError):
        hip.FdLib(str(tmp_path / "nope.so"))
    # the product loader only accepts the gfx950 backend
    assert emu_lib.backend == "emu" and not emu_lib.is_device


def test_error_reporting(emu_lib):
    import torch
    with pytest.raises(hip.FdError, match="multiple of 64"):
        emu_lib.call("fd_layernorm_fwd", torch.zeros(2, 100), 100, torch.ones(100), torch.zeros(100), None,
                     torch.zeros(2, 100), 100, None, None, 2, 100, 1e-5)


def test_probe_macros_do_not_compile_in_a_product_build(tmp_path):
    """a stray ablation macro (-DEM_ABLATE_PQ, -DFL_ABL_NOKV, ...) is a compile error unless the build is declared a probe build"""
    import subprocess
    from se3_diffusion_amd import build
    for src, macro in (("fd_edge_mlp.hip", "EM_ABLATE_PQ"), ("fd_ipa_flash.hip", "FL_ABL_NOKV")):
        cmd = [build.HIPCC, *build.FLAGS, f"-D{macro}", "-fsyntax-only", "--cuda-host-only", os.path.join(build.CSRC, src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode != 0 and "probe / ablation macro" in r.stderr, (src, r.stderr[-500:])
