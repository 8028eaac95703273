"""M8 (reference model/ipa_pytorch.py:23-86,101-166): the AF2 initialisers behind Linear(init=...) -- statistics pinned to the
reference's definitions (and to a freshly constructed reference model when /root/reference is present) -- and the on-disk
checkpoint format of the reference's du.write_checkpoint / read_pkl ({'model','conf','optimizer','epoch','step'},
data/utils.py:324-362,62-68) written and read back through THIS repo's ScoreNetwork + FlatAdam."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3_diffusion_amd import train_step as ts  # noqa: E402
from se3_diffusion_amd.model import ipa_pytorch  # noqa: E402
from se3_diffusion_amd.model.score_network import ScoreNetwork  # noqa: E402


def test_linear_init_modes():
    torch.manual_seed(0)
    fan_in, fan_out = 512, 384
    # default = LeCun truncated normal: std sqrt(1 / fan_in), truncated at +-2 sigma' (sigma' = std / 0.8796)
    for init, scale in (("default", 1.0), ("relu", 2.0)):
        w = ipa_pytorch.Linear(fan_in, fan_out, init=init)
        std = math.sqrt(scale / fan_in)
        assert abs(float(w.weight.std()) - std) < 0.02 * std, init
        assert float(w.weight.abs().max()) <= 2.0 * std / 0.87962566103423978 + 1e-7       # hard truncation
        assert float(w.bias.abs().max()) == 0.0
    w = ipa_pytorch.Linear(fan_in, fan_out, init="final")
    assert float(w.weight.abs().max()) == 0.0 and float(w.bias.abs().max()) == 0.0
    w = ipa_pytorch.Linear(fan_in, fan_out, init="gating")
    assert float(w.weight.abs().max()) == 0.0 and bool((w.bias == 1.0).all())
    w = ipa_pytorch.Linear(fan_in, fan_out, init="glorot")
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    assert float(w.weight.abs().max()) <= lim and abs(float(w.weight.std()) - lim / math.sqrt(3)) < 0.02 * lim
    w = ipa_pytorch.Linear(fan_in, fan_out, init="normal")
    assert abs(float(w.weight.std()) - math.sqrt(1.0 / fan_in)) < 0.02 * math.sqrt(1.0 / fan_in)
    with pytest.raises(ValueError):
        ipa_pytorch.Linear(4, 4, init="nope")


def test_model_init_follows_reference_choices():
    """final-initialised layers are exactly zero (so rot_score = psi = 0 at init), the IPA head weights are
    softplus^-1(1), every other Linear of the trunk has the std its init= mode prescribes."""
    torch.manual_seed(1)
    m = ScoreNetwork(ts.base_model_conf(2), diffuser=None)
    sd = m.state_dict()
    zero = ["score_model.trunk.node_transition_0.linear_3.weight", "score_model.trunk.bb_update_0.linear.weight",
            "score_model.trunk.ipa_0.linear_out.weight", "score_model.trunk.edge_transition_0.final_layer.weight",
            "score_model.trunk.post_tfmr_0.weight", "score_model.torsion_pred.linear_final.weight",
            "score_model.torsion_pred.linear_3.weight", "score_model.trunk.skip_embed_0.weight"]
    for k in zero:
        assert float(sd[k].abs().max()) == 0.0, k
    assert torch.allclose(sd["score_model.trunk.ipa_0.head_weights"], torch.full((8,), 0.541324854612918))
    he = ["score_model.trunk.node_transition_0.linear_1.weight", "score_model.trunk.edge_transition_0.trunk.0.weight",
          "score_model.trunk.edge_transition_0.initial_embed.weight", "score_model.torsion_pred.linear_1.weight",
          "score_model.torsion_pred.linear_2.weight"]
    for k in he:
        std = math.sqrt(2.0 / sd[k].shape[1])
        assert abs(float(sd[k].std()) - std) < 0.03 * std, k
    lecun = ["score_model.trunk.ipa_0.linear_q.weight", "score_model.trunk.ipa_0.linear_kv.weight",
             "score_model.trunk.ipa_0.linear_b.weight", "score_model.trunk.ipa_0.down_z.weight"]
    for k in lecun:
        std = math.sqrt(1.0 / sd[k].shape[1])
        assert abs(float(sd[k].std()) - std) < 0.06 * std, k


def test_init_statistics_match_reference_model():
    from oracle import ref_loader as rl
    if not rl.available():
        pytest.skip("reference not on this machine")
    import subprocess
    # the reference's package names (model, data) are generic: import it in its own process
    code = ("import sys, json, torch; sys.path.insert(0, %r); from oracle import ref_loader as rl; rl.install();"
            "from model import score_network; torch.manual_seed(0);"
            "m = score_network.ScoreNetwork(rl.base_conf('/tmp/fd_igso3_cache', num_blocks=2).model, None);"
            "print(json.dumps({k: [float(v.float().std()) if v.numel() > 1 else 0.0, float(v.float().mean())] "
            "for k, v in m.state_dict().items()}))") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    ref = json.loads(out.stdout.strip().splitlines()[-1])
    torch.manual_seed(0)
    mine = ScoreNetwork(ts.base_model_conf(2), diffuser=None).state_dict()
    assert set(ref) == set(mine)
    bad = []
    for k, (std, mean) in ref.items():
        v = mine[k].float()
        n = v.numel()
        s = float(v.std()) if n > 1 else 0.0
        tol = 6.0 / math.sqrt(max(n, 2)) + 0.01            # sampling error of a std estimate, ~ 1 / sqrt(2 n)
        if abs(s - std) > tol * max(std, 1e-12) + 1e-9 or abs(float(v.mean()) - mean) > 6.0 * max(std, s) / math.sqrt(n) + 1e-7:
            bad.append((k, s, std, float(v.mean()), mean))
    assert not bad, bad[:8]


def test_checkpoint_round_trip_reference_layout(use_emu, tmp_path):
    """write_checkpoint's dict layout (data/utils.py:353-362): {'model': state_dict, 'conf', 'optimizer': Adam
    state_dict, 'epoch', 'step'} pickled with torch.save (write_pkl use_torch=True, :62-68) -- written from
    ScoreNetwork + FlatAdam after a step, read back, loaded with strict=True into a fresh model and into
    torch.optim.Adam (the reference's optimiser, train_se3_diffusion.py:139-142), and training continues identically."""
    from oracle import framediff_oracle as fo
    from se3_diffusion_amd.optim import FlatAdam
    conf = dict(fo.CONF, num_blocks=1)
    model = ScoreNetwork(ts.base_model_conf(1), diffuser=None)
    model.load_state_dict(fo.synth_params(seed=5, conf=conf), strict=True)
    model.train()
    batch = ts.synthetic_batch(1, 8, "cpu", seed=2)
    gt37, _ = fo.backbone_atoms(batch["rigids_0"][..., :4], batch["rigids_0"][..., 4:], batch["torsion_angles_sin_cos"][..., 2, :])
    opt = FlatAdam(model.parameters(), lr=1e-4)

    def one_step(m, o):
        o.zero_grad()
        ts.dsm_loss(batch, m(batch), gt37).backward()
        o.step()

    one_step(model, opt)
    path = str(tmp_path / "step_1.pth")
    torch.save({"model": model.state_dict(), "conf": {"model": {"ipa": {"num_blocks": 1}}}, "optimizer": opt.state_dict(),
                "epoch": 0, "step": 1}, path, pickle_protocol=4)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ckpt) == {"model", "conf", "optimizer", "epoch", "step"}
    # the reference strips a DataParallel 'module.' prefix before loading (train_se3_diffusion.py:133-134): none here
    assert not any(k.startswith("module.") for k in ckpt["model"])
    m2 = ScoreNetwork(ts.base_model_conf(1), diffuser=None)
    m2.load_state_dict(ckpt["model"], strict=True)
    m2.train()
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-4)              # the reference's optimiser class
    o2.load_state_dict(ckpt["optimizer"])
    m3 = ScoreNetwork(ts.base_model_conf(1), diffuser=None)
    m3.load_state_dict(ckpt["model"], strict=True)
    m3.train()
    o3 = FlatAdam(m3.parameters(), lr=1e-4)
    o3.load_state_dict(ckpt["optimizer"])
    one_step(model, opt)
    one_step(m2, o2)
    one_step(m3, o3)
    for (n, a), b, c in zip(model.named_parameters(), m2.parameters(), m3.parameters()):
        assert float((a - b).abs().max()) <= 2e-7 * (1 + float(a.abs().max())), n     # torch Adam continues FlatAdam's state
        assert torch.equal(a.detach(), c.detach()), n                                  # FlatAdam resumes bit-identically
