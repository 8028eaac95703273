"""What the weight-gradient side stream costs the training step: bench.py's timed step with the fd_group_dw and / or fd_pair_dw
launches left out (WRONG gradients by design -- timing only; nothing in the product reads these switches).
   SKIP=group|pair|both|zb python tools/probes/skip_dw.py --steps 30 --warmup 8 --no-cpu-baseline --no-sampling   (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from se3_diffusion_amd import ops  # noqa: E402

skip = os.environ.get("SKIP", "")
if skip in ("group", "both"):
    ops.group_dw = lambda *a, **k: None
if skip in ("pair", "both"):
    ops.pair_dw = lambda *a, **k: None
if skip == "zb":          # the [40, 128] weight gradient of IPA's pair bias / down_z (+ its bias): 4 fp32 split-K GEMMs + 4 column sums per step
    _side = ops.side
    ops.side = lambda fn, *a, **k: None if getattr(fn, "__name__", "") == "_grads_zb" else _side(fn, *a, **k)
bench.main()
