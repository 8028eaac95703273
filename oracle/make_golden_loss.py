"""TEST INFRASTRUCTURE -- golden vectors for the training loss from the UNMODIFIED reference
(experiments/train_se3_diffusion.py:524-693 Experiment.loss_fn), both rotation-loss branches
(separate_rot_loss=True of config/base.yaml, =False of config/icml_published.yaml).

Run in the build container only (needs /root/reference):
    python oracle/make_golden_loss.py
Writes tests/golden/loss.npz: the batch, fixed "network outputs", and for each branch the loss, the per-example
terms the reference logs, and the gradient of the loss w.r.t. the outputs.  Experiment.__init__ (data loaders, wandb,
checkpoint directories) is bypassed: loss_fn only reads the three config nodes and calls self.model(batch).
"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader as rl  # noqa: E402
from oracle.make_golden import GOLD  # noqa: E402


def main():
    rl.install()
    from hydra.core.hydra_config import HydraConfig
    HydraConfig.initialized = lambda: False
    from experiments import train_se3_diffusion as tr
    from se3_diffusion_amd import train_step as ts          # synthetic batch generator only

    B, N = 3, 9
    batch = ts.synthetic_batch(B, N, "cpu", seed=1)
    batch["t"] = torch.tensor([0.1, 0.22, 0.9])               # both sides of the 0.2 / 0.25 filters
    batch["res_mask"][:, N - 2:] = 0
    batch["fixed_mask"][:, :1] = 1
    g = torch.Generator().manual_seed(7)
    outs = dict(rot_score=torch.randn(B, N, 3, generator=g, dtype=torch.float64),
                trans_score=torch.randn(B, N, 3, generator=g),
                rigids=torch.cat([torch.randn(B, N, 4, generator=g), batch["rigids_0"][..., 4:] + torch.randn(B, N, 3, generator=g)], -1),
                atom37=torch.randn(B, N, 37, 3, generator=g) * 3)
    d = {"batch/" + k: v.numpy() for k, v in batch.items()}
    d.update({"out/" + k: v.numpy() for k, v in outs.items()})
    from data import all_atom
    from openfold.utils import rigid_utils as ru
    gt37 = all_atom.compute_backbone(ru.Rigid.from_tensor_7(batch["rigids_0"].float()), batch["torsion_angles_sin_cos"][..., 2, :])[0]
    d["gt_atom37"] = gt37.numpy()
    exp_conf = dict(trans_loss_weight=1.0, rot_loss_weight=0.5, rot_loss_t_threshold=0.2, trans_x0_threshold=1.0,
                    coordinate_scaling=0.1, bb_atom_loss_weight=1.0, bb_atom_loss_t_filter=0.25,
                    dist_mat_loss_weight=1.0, dist_mat_loss_t_filter=0.25, aux_loss_weight=0.25)
    for tag, sep in (("sep", True), ("joint", False)):
        exp = object.__new__(tr.Experiment)
        exp._exp_conf = rl.ns(dict(exp_conf, separate_rot_loss=sep))
        exp._diff_conf = rl.ns(dict(diffuse_rot=True, diffuse_trans=True))
        exp._model_conf = rl.ns(dict(embed=dict(embed_self_conditioning=False)))
        exp._aux_data_history = collections.deque(maxlen=4)
        o = {k: v.clone().requires_grad_(True) for k, v in outs.items()}
        exp._model = lambda b, o=o: o
        loss, aux = exp.loss_fn({k: v.clone() for k, v in batch.items()})
        loss.backward()
        d[f"{tag}/loss"] = float(loss)
        for k in ("batch_train_loss", "batch_rot_loss", "batch_trans_loss", "batch_bb_atom_loss", "batch_dist_mat_loss"):
            d[f"{tag}/{k}"] = aux[k].detach().numpy()
        for k, v in o.items():
            d[f"{tag}/grad/{k}"] = v.grad.numpy()
        print(tag, float(loss), aux["batch_rot_loss"].detach().numpy(), flush=True)
    np.savez_compressed(os.path.join(GOLD, "loss.npz"), **d)
    print("wrote", os.path.join(GOLD, "loss.npz"))


if __name__ == "__main__":
    main()
