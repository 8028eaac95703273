"""Which Python lines issue the device-to-device copies of a training step?  Wraps Tensor.copy_ / clone / contiguous / to for one
step and prints the call sites with counts.   python tools/copy_trace.py   (GPU box)"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from se3_diffusion_amd import loss as floss, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    from se3_diffusion_amd.optim import FlatAdam
    dev = "cuda"
    B, N = 30, 128
    model = ScoreNetwork(ts.base_model_conf(4), diffuser=None).to(dev)
    ts.perturb_final_layers(model, seed=0)
    model.train()
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=model.flat_layout_groups())
    model.accumulate_into_grad = True
    batch = ts.synthetic_batch(B, N, dev, seed=100)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])

    def step():
        opt.zero()
        out = model(batch)
        loss = floss.dsm_loss(batch, out, gt37)
        loss.backward()
        opt.all_reduce_mean()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    sites = collections.Counter()
    orig = {}

    def wrap(name):
        f = getattr(torch.Tensor, name)
        orig[name] = f

        def g(self, *a, **k):
            if self.is_cuda:
                fr = [x for x in traceback.extract_stack()[:-1] if "se3_diffusion_amd" in x.filename or "bench" in x.filename]
                if fr:
                    x = fr[-1]
                    changed = True
                    if name == "contiguous":
                        changed = not self.is_contiguous()
                    if name == "to":
                        changed = None
                    sites[(name, os.path.basename(x.filename), x.lineno, x.line.strip()[:90], changed)] += 1
            return f(self, *a, **k)
        setattr(torch.Tensor, name, g)

    for n in ("copy_", "clone", "contiguous", "to", "zero_", "fill_", "float", "double"):
        wrap(n)
    step()
    torch.cuda.synchronize()
    for n, f in orig.items():
        setattr(torch.Tensor, n, f)
    for k, v in sorted(sites.items(), key=lambda kv: -kv[1]):
        if k[0] == "contiguous" and k[4] is False:
            continue
        print(v, k)


if __name__ == "__main__":
    main()
