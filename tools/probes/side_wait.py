"""Does pair-level work on BOTH streams cost more than it hides?  bench.py's step with the main stream waiting for the gradient side
stream in front of every fused edge-transition backward (so fd_pair_dw / fd_group_dw never run beside the next pair-level MFMA kernel)
and / or in front of the edge embedder's backward.   WAIT=edge|embed|both python tools/probes/side_wait.py <bench.py arguments>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from se3_diffusion_amd import network, ops, trunk  # noqa: E402

mode = os.environ.get("WAIT", "")


def _wait():
    cur = torch.cuda.current_stream()
    for st in ops._SIDE["streams"].values():
        cur.wait_stream(st)


if mode in ("edge", "both"):
    _etb = trunk.edge_transition_bwd

    def etb(*a, **k):
        _wait()
        return _etb(*a, **k)
    trunk.edge_transition_bwd = etb
if mode in ("embed", "both"):
    _eb = network.embed_bwd if hasattr(network, "embed_bwd") else None
    assert _eb is not None, "network.embed_bwd not found"

    def eb(*a, **k):
        _wait()
        return _eb(*a, **k)
    network.embed_bwd = eb
bench.main()
