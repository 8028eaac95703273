"""pytest configuration.

Markers:
  gpu -- needs a real MI355X (driver runs `-m gpu` on the GPU box; `-m "not gpu"` here).

Fixtures:
  emu_lib -- the kernel sources compiled for the host SIMT interpreter
             (tests/emu, TEST INFRASTRUCTURE) so kernel logic is checked on CPU.
  hip_lib -- the product gfx950 library on a GPU.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real AMD GPU (MI355X)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def emu_lib():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from se3_diffusion_amd import hip
    path = build_emu.build(verbose=False)
    lib = hip.FdLib(path)
    assert lib.backend == "emu"
    return lib


@pytest.fixture()
def use_emu(emu_lib):
    """Route se3_diffusion_amd.hip.get_lib() to the interpreter for one test."""
    from se3_diffusion_amd import hip
    hip._TEST_OVERRIDE = emu_lib
    yield emu_lib
    hip._TEST_OVERRIDE = None


@pytest.fixture(scope="session")
def hip_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from se3_diffusion_amd import hip
    return hip.get_lib()
