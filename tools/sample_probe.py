"""One sampling configuration with the sampler's own statistics: ms per diffusion step (device time of the replayed loop) and the
library's kernels per captured step.   python tools/sample_probe.py N B [num_t]   (GPU box; options through FD_* variables)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from se3_diffusion_amd import sampler, train_step as ts  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402

N, B = int(sys.argv[1]), int(sys.argv[2])
num_t = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = "cuda"
diff, _ = bench.make_diffuser()
torch.manual_seed(0)
model = ScoreNetwork(ts.base_model_conf(4), diff).to(dev)
ts.perturb_final_layers(model, seed=0)
model.eval()
gen = torch.Generator(device=dev).manual_seed(99)
res = []
for rep in range(3):
    st = {}
    feats = sampler.init_feats(diff, B, N, dev, generator=gen)
    sampler.sample(model, diff, feats, num_t=num_t, min_t=0.01, noise_scale=0.1, generator=gen, use_graph=True, stats=st)
    torch.cuda.synchronize()
    res.append(st["loop_ms"] / st["steps"])
print(f"N={N} B={B}: ms per diffusion step {min(res[1:]):.4f} (runs {[round(r, 4) for r in res]}), kernels per captured step {st.get('kernels_per_step')}, "
      f"backbones/s at 501 forwards {B / (min(res[1:]) * 501e-3):.3f}")
