"""GEMM launches of one training step by shape: count, us per launch (HIP events around every launch, serialised), total.
    python tools/gemm_shapes.py [B N]      (GPU box)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from se3_diffusion_amd import hip, loss as floss, options, train_step as ts
    from se3_diffusion_amd.model.score_network import ScoreNetwork
    from se3_diffusion_amd.optim import FlatAdam
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = "cuda"
    model = ScoreNetwork(ts.base_model_conf(4), diffuser=None).to(dev)
    ts.perturb_final_layers(model, seed=0)
    model.train()
    opt = FlatAdam(model.parameters(), lr=1e-4, adjacent=model.flat_layout_groups())
    model.accumulate_into_grad = True
    batch = ts.synthetic_batch(B, N, dev, seed=100)
    gt37, _ = ts.backbone_atoms(batch["rigids_0"], batch["torsion_angles_sin_cos"][..., 2, :])

    def step():
        opt.zero()
        loss = floss.dsm_loss(batch, model(batch), gt37)
        loss.backward()
        opt.step()

    with options.override(grad_stream=False):
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lib = hip.get_lib()
        lib.gemm_profile = []
        step()
        torch.cuda.synchronize()
        recs, lib.gemm_profile = lib.gemm_profile, None
    agg = collections.defaultdict(lambda: [0, 0.0])
    for tile, a_kc, b_kc, flops, e0, e1, shape in recs:
        k = (tile, a_kc, b_kc) + shape
        agg[k][0] += 1
        agg[k][1] += e0.elapsed_time(e1) * 1e3
    tot = sum(v[1] for v in agg.values())
    print(f"B={B} N={N}: {len(recs)} fd_gemm launches, {tot / 1e3:.2f} ms (events around each launch, side stream off)")
    print("   n   us/launch   total us  TFLOP/s  tile a_kc b_kc  (M, N, K, batch, gate, beta, pair, ksplit)")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        M, Nn, K, bt = k[3], k[4], k[5], k[6]
        print(f"{n:4d}  {t / n:9.1f}  {t:9.1f}  {2.0 * M * Nn * K * bt * n / t / 1e6:7.1f}  {k[0]:4d} {int(k[1]):4d} {int(k[2]):4d}  {k[3:]}")


if __name__ == "__main__":
    main()
