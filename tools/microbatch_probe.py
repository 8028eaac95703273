"""Probe: two half-batch training steps (B = 15 each, fwd + loss + bwd) replayed from two hipGraphs on two streams at
the same time, against one B = 30 step -- does interleaving two micro-batches let the latency-bound node-level launches
of one overlap the pair-level kernels of the other?  Timing experiment only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import loss as floss, ops, train_step as ts  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402
from se3_diffusion_amd.optim import FlatAdam  # noqa: E402
import bench  # noqa: E402


def make(B, N, dev, seed):
    diff, _ = bench.make_diffuser()
    torch.manual_seed(0)
    model = ScoreNetwork(ts.base_model_conf(4), diff).to(dev)
    ts.perturb_final_layers(model, seed=0)
    model.train()
    model.accumulate_into_grad = True
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=model.flat_layout_groups())
    batch = ts.synthetic_batch(B, N, dev, seed=seed)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])

    def step():
        opt.zero()
        out = model(batch)
        floss.dsm_loss(batch, out, gt37).backward()
    return step


def capture(step):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    return g


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    side = os.environ.get("PROBE_SIDE", "0") == "1"
    ops.set_grad_stream(side)
    full = make(30, 128, dev, 100)
    print(f"eager B=30 (side stream {side}): {timeit(full):.2f} ms", flush=True)
    g30 = capture(full)
    print(f"graph B=30: {timeit(g30.replay):.2f} ms", flush=True)
    ha, hb = make(15, 128, dev, 101), make(15, 128, dev, 102)
    ga, gb = capture(ha), capture(hb)
    print(f"graph B=15 alone: {timeit(ga.replay):.2f} ms", flush=True)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s1):
            ga.replay()
        with torch.cuda.stream(s2):
            gb.replay()
    print(f"two B=15 graphs on two streams: {timeit(both):.2f} ms per pair", flush=True)


if __name__ == "__main__":
    main()
