"""Make the reference's own scripts run on the MI355X hot path, unchanged.

The reference has no plugin/FFI registry: experiments/train_se3_diffusion.py and
experiments/inference_se3_diffusion.py locate the hot path purely by MODULE PATH
(`from model import score_network`, `from data import se3_diffuser`,
`from openfold.utils import rigid_utils as ru`, train_se3_diffusion.py:38-47).  `install()` puts
the reference checkout on sys.path (its packages are namespace packages) and binds those module
paths to this package's implementations, so every other reference module (data loaders, analysis,
experiment utils, configs) keeps working as is:

    import se3_diffusion_amd.dropin as dropin
    dropin.install("/path/to/se3_diffusion")          # before importing experiments.*
    from experiments import train_se3_diffusion       # now trains through libfd_hip.so

or, from a shell:  python -m se3_diffusion_amd.dropin /path/to/se3_diffusion experiments/train_se3_diffusion.py [hydra args]
"""
from __future__ import annotations

import importlib
import os
import runpy
import sys

_BINDINGS = {
    "model.score_network": "se3_diffusion_amd.model.score_network",
    "model.ipa_pytorch": "se3_diffusion_amd.model.ipa_pytorch",
    "data.se3_diffuser": "se3_diffusion_amd.data.se3_diffuser",
    "data.so3_diffuser": "se3_diffusion_amd.data.so3_diffuser",
    "data.r3_diffuser": "se3_diffusion_amd.data.r3_diffuser",
}
_RIGID = {"openfold.utils.rigid_utils": "se3_diffusion_amd.openfold.utils.rigid_utils"}


def install(reference_root: str | None = None, replace_rigid_utils: bool = False, patch_all_atom: bool = True):
    """Bind the reference's hot-path module paths to the HIP implementations.  Idempotent.

    replace_rigid_utils: also bind openfold.utils.rigid_utils to the eigh-free Rigid/Rotation of this
    package (only the subset of SURVEY.md 8a M14 is implemented; the reference's offline data
    pipeline uses more of that file, so the default keeps the reference's)."""
    if reference_root is not None:
        reference_root = os.path.abspath(reference_root)
        if not os.path.isdir(os.path.join(reference_root, "experiments")):
            raise FileNotFoundError(f"{reference_root} does not look like a se3_diffusion checkout")
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)
    table = dict(_BINDINGS)
    if replace_rigid_utils:
        table.update(_RIGID)
    for ref_name, our_name in table.items():
        mod = importlib.import_module(our_name)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition(".")
        try:
            setattr(importlib.import_module(parent), leaf, mod)
        except ImportError:
            pass   # parent package not importable (no reference on sys.path): the sys.modules entry is enough
    if patch_all_atom:
        try:
            ref_all_atom = importlib.import_module("data.all_atom")
            ours = importlib.import_module("se3_diffusion_amd.data.all_atom")
            ref_all_atom.compute_backbone = ours.compute_backbone     # kills the per-call host round trip
        except ImportError:
            sys.modules.setdefault("data.all_atom", importlib.import_module("se3_diffusion_amd.data.all_atom"))
    return sorted(table)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 2:
        raise SystemExit("usage: python -m se3_diffusion_amd.dropin <reference_root> <script.py> [args...]")
    root, script = argv[0], argv[1]
    install(root)
    sys.argv = [script] + argv[2:]
    os.chdir(root)
    runpy.run_path(os.path.join(root, script) if not os.path.isabs(script) else script, run_name="__main__")


if __name__ == "__main__":
    main()
