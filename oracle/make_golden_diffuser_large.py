"""TEST INFRASTRUCTURE -- the diffuser entry points at the sizes the product ships at, from the UNMODIFIED reference
(data/se3_diffuser.py:160-268): tests/golden/diffuser.npz checks them at n = 9..11 residues; this fixture adds
    reverse            B=3 x N=300 with fixed residues, t = 0.6 / 0.03 (se3_diffuser.py:160-214)
    sample_ref         700 residues (216-268)
    forward_marginal   N=300 at t = 0.37 (43-110)
with the reference's numpy stream: only the SEEDS are stored (legacy numpy streams are stable across versions), the test
re-draws the normal / uniform variates in the reference's call order.

Run in the build container only (needs /root/reference):   python oracle/make_golden_diffuser_large.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader as rl  # noqa: E402
from oracle import framediff_oracle as fo  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CACHE = os.environ.get("FD_IGSO3_CACHE", "/tmp/fd_igso3_cache")


def inputs():
    """the seeded inputs, shared with tests/test_diffuser.py"""
    Bn, Nn = 3, 300
    rs = np.random.RandomState(77)
    rig = fo.synth_feats(Bn, Nn, seed=19)["rigids_t"]
    rot_score = rs.standard_normal((Bn, Nn, 3)) * 0.5
    trans_score = rs.standard_normal((Bn, Nn, 3)) * 0.5
    dmask = (rs.uniform(size=(Bn, Nn)) > 0.1).astype(np.float64)
    rig0 = fo.synth_feats(1, Nn, seed=23)["rigids_t"][0]
    return dict(B=Bn, N=Nn, rig=rig, rot_score=rot_score, trans_score=trans_score, dmask=dmask, rig0=rig0)


def main():
    rl.install()
    from data import se3_diffuser
    from openfold.utils import rigid_utils as ru
    diff = se3_diffuser.SE3Diffuser(rl.base_conf(CACHE).diffuser)
    x = inputs()
    d = dict(B=x["B"], N=x["N"])
    for tag, t_, ns_, seed in (("a", 0.6, 1.0, 1123), ("b", 0.03, 0.1, 1124)):
        np.random.seed(seed)
        out = diff.reverse(rigid_t=ru.Rigid.from_tensor_7(x["rig"].clone()), rot_score=x["rot_score"], trans_score=x["trans_score"],
                           t=t_, dt=1 / 500, diffuse_mask=x["dmask"], center=True, noise_scale=ns_)
        d[f"rev_{tag}_t"], d[f"rev_{tag}_ns"], d[f"rev_{tag}_seed"] = t_, ns_, seed
        d[f"rev_{tag}_out_t7"] = out.to_tensor_7().numpy().astype(np.float32)
    np.random.seed(1321)
    d["sr_seed"], d["sr_n"] = 1321, 700
    d["sr_out_t7"] = diff.sample_ref(n_samples=700, as_tensor_7=True)["rigids_t"].numpy().astype(np.float32)
    np.random.seed(155)
    fm = diff.forward_marginal(ru.Rigid.from_tensor_7(x["rig0"]), 0.37, diffuse_mask=None, as_tensor_7=True)
    d["fm_seed"], d["fm_t"] = 155, 0.37
    d["fm_rigids_t"] = fm["rigids_t"].numpy().astype(np.float32)
    d["fm_trans_score"], d["fm_rot_score"] = np.asarray(fm["trans_score"]), np.asarray(fm["rot_score"])
    np.savez_compressed(os.path.join(GOLD, "diffuser_large.npz"), **d)
    print("wrote diffuser_large.npz", {k: getattr(v, "shape", v) for k, v in d.items()})


if __name__ == "__main__":
    main()
