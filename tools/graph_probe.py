"""Probe: training step (fwd + loss + bwd + Adam) replayed from one hipGraph vs eager launches.  Timing experiment only
(the captured Adam step keeps the bias corrections of the capture step)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import hip, loss as floss, ops, train_step as ts  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402
from se3_diffusion_amd.optim import FlatAdam  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, N = 30, 128
    diff, _ = bench.make_diffuser()
    torch.manual_seed(0)
    model = ScoreNetwork(ts.base_model_conf(4), diff).to(dev)
    ts.perturb_final_layers(model, seed=0)
    model.train()
    model.accumulate_into_grad = True
    opt = FlatAdam(model.parameters(), lr=1e-4)
    batch = ts.synthetic_batch(B, N, dev, seed=100)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])

    def step():
        opt.zero()
        out = model(batch)
        l = floss.dsm_loss(batch, out, gt37)
        l.backward()
        opt.step()

    def timeit(fn, n=10):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print(f"eager            {timeit(step):.2f} ms", flush=True)
    ops.set_grad_stream(False)
    print(f"eager, 1 stream  {timeit(step):.2f} ms", flush=True)
    for side in (False, True):
        ops.set_grad_stream(side)
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
        print(f"graph replay, side stream {side}: {timeit(g.replay):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
