"""cProfile of the host side of a training step (B=30 x N=128): where the ~12 ms of enqueue time per step go."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se3_diffusion_amd import loss as floss, train_step as ts  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402
from se3_diffusion_amd.optim import FlatAdam  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    diff, _ = bench.make_diffuser()
    torch.manual_seed(0)
    model = ScoreNetwork(ts.base_model_conf(4), diff).to(dev)
    ts.perturb_final_layers(model, seed=0)
    model.train()
    model.accumulate_into_grad = True
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=model.flat_layout_groups())
    batch = ts.synthetic_batch(30, 128, dev, seed=100)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])

    def step():
        opt.zero()
        out = model(batch)
        floss.dsm_loss(batch, out, gt37).backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
