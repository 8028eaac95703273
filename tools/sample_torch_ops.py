"""Which Python lines launch the non-library (torch / runtime) kernels of a sampling step?  Runs a few eager reverse steps under
torch.profiler with stacks and prints, per device kernel that is NOT one of libfd_hip.so's, the launching call site inside
se3_diffusion_amd/ with its count per step.      python tools/sample_torch_ops.py [N [B]]      (GPU box)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from se3_diffusion_amd import sampler, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    import bench
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dev = "cuda"
    diffuser, _ = bench.make_diffuser()
    model = ScoreNetwork(ts.base_model_conf(4), diffuser=diffuser).to(dev).eval()
    ts.perturb_final_layers(model, seed=0)
    feats = sampler.init_feats(diffuser, B, N, dev)
    steps = 6
    sampler.sample(model, diffuser, feats, num_t=steps, use_graph=False)       # warm
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        sampler.sample(model, diffuser, feats, num_t=steps, use_graph=False)
        torch.cuda.synchronize()
    sites = collections.Counter()
    names = collections.defaultdict(collections.Counter)
    for ev in prof.events():
        if ev.device_type.name != "CPU" or not ev.kernels:
            continue
        stack = [s for s in (ev.stack or []) if "se3_diffusion_amd" in s or "sampler" in s]
        site = stack[0].strip() if stack else "(no frame inside the package)"
        for k in ev.kernels:
            sites[site] += 1
            names[site][k.name[:70]] += 1
    print(f"N={N} B={B}: torch-launched device kernels over {steps} steps (+1 self-conditioning forward), by call site")
    for site, c in sites.most_common(40):
        print(f"{c:5d}  {site}")
        for k, n in names[site].most_common(3):
            print(f"         {n:4d} x {k}")


if __name__ == "__main__":
    main()
