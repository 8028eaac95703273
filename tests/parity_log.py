"""TEST INFRASTRUCTURE -- achieved-error bookkeeping of the parity tests.

Every parity case runs inside `with parity_log.case("name"):`; the output / gradient checks report what they MEASURED
(`out(key, err)`, `grad(name, g, g_ref)`), and on exit the case prints one `[parity] ...` line (pytest -s, or the failure
text) and appends a JSON record to FD_PARITY_LOG (default gpurun_out/parity_errors.jsonl), so that the bounds written in the
tests can be compared with what the code achieves -- and drift shows up as numbers, not as a pass that got closer to failing.

Gradient classes = the parameter families of the state_dict.  Two figures per gradient tensor:
  maxrel = max |g - g_ref| / max |g_ref|          (what the bounds in the tests are written against)
  l2rel  = ||g - g_ref||_2 / ||g_ref||_2          (a wrong small-magnitude region -- a bias row, a rarely-hit tile tail -- that
                                                   hides under a max-norm bound shows here)
"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CUR = None

_FAMILIES = (
    (r"embedding_layer\.node_embedder", "embed.node"), (r"embedding_layer\.edge_embedder", "embed.edge"),
    (r"embedding_layer", "embed.other"),
    (r"trunk\.ipa_\d+\.(linear_b|down_z)", "ipa.pair_proj"), (r"trunk\.ipa_\d+\.head_weights", "ipa.head_weights"),
    (r"trunk\.ipa_\d+", "ipa.proj"), (r"trunk\.ipa_ln_\d+", "ipa_ln"), (r"trunk\.skip_embed_\d+", "skip_embed"),
    (r"trunk\.seq_tfmr_\d+", "seq_tfmr"), (r"trunk\.post_tfmr_\d+", "post_tfmr"),
    (r"trunk\.node_transition_\d+", "node_transition"), (r"trunk\.bb_update_\d+", "bb_update"),
    (r"trunk\.edge_transition_\d+", "edge_transition"), (r"torsion_pred", "torsion_pred"),
)


def family(name):
    for pat, fam in _FAMILIES:
        if re.search(pat, name):
            return fam
    return "other"


class case:
    def __init__(self, name):
        self.name = name
        self.outs, self.gmax, self.gl2, self.kinks = {}, {}, {}, 0

    def __enter__(self):
        global _CUR
        self.prev, _CUR = _CUR, self
        return self

    def __exit__(self, et, ev, tb):
        global _CUR
        _CUR = self.prev
        rec = {"case": self.name, "passed": et is None, "outputs": {k: float(f"{v:.3e}") for k, v in self.outs.items()},
               "grad_maxrel": {k: [float(f"{v[0]:.3e}"), v[1]] for k, v in self.gmax.items()},
               "grad_l2rel": {k: [float(f"{v[0]:.3e}"), v[1]] for k, v in self.gl2.items()}}
        worst_m = max(self.gmax.values(), default=(0.0, ""))
        worst_l = max(self.gl2.values(), default=(0.0, ""))
        print(f"[parity] {self.name}: outputs {rec['outputs']} | grads worst maxrel {worst_m[0]:.2e} ({worst_m[1]}) "
              f"worst l2rel {worst_l[0]:.2e} ({worst_l[1]}) | by family maxrel "
              f"{ {k: float(f'{v[0]:.1e}') for k, v in sorted(self.gmax.items())} }")
        path = os.environ.get("FD_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl"))
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "a") as f:
                f.write(json.dumps(rec) + "\n")
        except OSError:
            pass
        return False


def out(key, err):
    if _CUR is not None:
        _CUR.outs[key] = max(_CUR.outs.get(key, 0.0), float(err))


def grad(name, g, g_ref, excused=False):
    """record maxrel / l2rel of one gradient tensor (float64 CPU tensors) under its parameter family; tensors whose reference
    is (analytically) zero are skipped -- their check is the absolute floor; entries excused as ReLU kinks are counted apart"""
    if _CUR is None or name is None:
        return
    scale = float(g_ref.abs().max())
    if scale < 1e-7 or name.endswith("linear_b.bias"):      # (linear_b.bias: analytically zero -- softmax shift invariance)
        return
    if excused:
        _CUR.kinks += 1
        return
    fam = family(name)
    d = g - g_ref
    m = float(d.abs().max()) / scale
    l2 = float(d.norm()) / (float(g_ref.norm()) + 1e-30)
    if m > _CUR.gmax.get(fam, (0.0, ""))[0]:
        _CUR.gmax[fam] = (m, name)
    if l2 > _CUR.gl2.get(fam, (0.0, ""))[0]:
        _CUR.gl2[fam] = (l2, name)
