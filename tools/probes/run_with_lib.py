"""Run a tool script with a variant library in place of the product one (measurements only):
    python tools/probes/run_with_lib.py tools/probes/libfd_var_TAG.so tools/bench_group_dw.py [args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import hip  # noqa: E402

lib = hip.FdLib(os.path.abspath(sys.argv[1]))
assert lib.backend == "gfx950"
hip._PRODUCT = lib
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
