"""TEST INFRASTRUCTURE -- import the *unmodified* reference (jasonkyuyim/se3_diffusion)
read-only from /root/reference, stubbing its missing non-arithmetic dependencies.

Only usable in the build container (the GPU box has no /root/reference).  Used by
oracle/make_golden.py to pin oracle/framediff_oracle.py and to write tests/golden/.
Recipe: SURVEY.md Appendix C.  Must run in its own process: the reference's
top-level package names (`model`, `data`, `openfold`) are generic.
"""
import importlib.machinery as im
import os
import sys
import types
from unittest import mock

REF_ROOT = os.environ.get("FD_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


def _stub(name, obj=None):
    m = obj or mock.MagicMock()
    m.__spec__ = im.ModuleSpec(name, None)
    sys.modules[name] = m
    return m


def _map_structure(fn, *s):
    if isinstance(s[0], dict):
        return {k: _map_structure(fn, *[x[k] for x in s]) for k in s[0]}
    if isinstance(s[0], (list, tuple)):
        return type(s[0])(_map_structure(fn, *xs) for xs in zip(*s))
    return fn(*s)


def install():
    """Put the reference on sys.path with stubs; idempotent."""
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "tree" not in sys.modules:
        tree = types.ModuleType("tree")
        tree.map_structure = _map_structure
        _stub("tree", tree)
    for n in ["omegaconf", "Bio", "Bio.PDB", "Bio.PDB.Chain", "Bio.Data", "ml_collections", "absl",
              "GPUtil", "wandb", "hydra", "hydra.core", "hydra.core.hydra_config", "mdtraj", "tmtools",
              "openfold.np.relax", "openfold.np.relax.amber_minimize", "dm_tree", "biotite", "esm"]:
        if n not in sys.modules:
            _stub(n)
    if "openfold.config" not in sys.modules:
        cfg = types.ModuleType("openfold.config")
        cfg.NUM_RES, cfg.NUM_MSA_SEQ, cfg.NUM_EXTRA_SEQ, cfg.NUM_TEMPLATES = "r", "m", "e", "t"
        _stub("openfold.config", cfg)


def ns(d):
    """dict tree -> attribute namespace tree (stands in for OmegaConf nodes)."""
    if isinstance(d, dict):
        return types.SimpleNamespace(**{k: ns(v) for k, v in d.items()})
    return d


def base_conf(cache_dir, num_blocks=4, use_cached_score=False):
    """config/base.yaml (reference config/base.yaml:25-67) as plain namespaces."""
    diffuser = dict(
        diffuse_trans=True, diffuse_rot=True,
        r3=dict(min_b=0.1, max_b=20.0, coordinate_scaling=0.1),
        so3=dict(num_omega=1000, num_sigma=1000, min_sigma=0.1, max_sigma=1.5,
                 schedule="logarithmic", cache_dir=cache_dir, use_cached_score=use_cached_score),
    )
    model = dict(
        node_embed_size=256, edge_embed_size=128, dropout=0.0,
        embed=dict(index_embed_size=32, aatype_embed_size=64, embed_self_conditioning=True,
                   num_bins=22, min_bin=1e-5, max_bin=20.0),
        ipa=dict(c_s=256, c_z=128, c_hidden=256, c_skip=64, no_heads=8, no_qk_points=8,
                 no_v_points=12, seq_tfmr_num_heads=4, seq_tfmr_num_layers=2,
                 num_blocks=num_blocks, coordinate_scaling=0.1),
    )
    return ns(dict(diffuser=diffuser, model=model))
