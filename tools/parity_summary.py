"""profiles/rNN_parity_errors.md from gpurun_out/parity_errors.jsonl (written by tests/parity_log.py during `pytest -m gpu`):
what every parity case ACHIEVED, next to the bounds written in tests/test_network.py.   python tools/parity_summary.py [out.md]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, "gpurun_out", "parity_errors.jsonl")
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_parity_errors.md")
    recs = [json.loads(l) for l in open(src)]
    for r in recs:      # linear_b.bias is analytically zero (softmax shift invariance): its check is the absolute floor, not a ratio
        for tab in ("grad_maxrel", "grad_l2rel"):
            r[tab] = {f: v for f, v in r[tab].items() if not v[1].endswith("linear_b.bias")}
    keys = ["psi", "rot_score", "trans_score", "atom37", "rigids"]
    fams = sorted({f for r in recs for f in r["grad_maxrel"]})
    L = [f"# Achieved parity errors, `pytest -m gpu` on MI355X ({len(recs)} cases; tests/parity_log.py)", "",
         "Outputs: max |got - ref| / max |ref| per tensor.  Gradients: worst tensor of the case, max-norm relative to the tensor's maximum",
         "(`maxrel`) and relative L2 (`l2rel`).  Bounds: tests/test_network.py (`TOL_OUT_KEY`, `TOL_GRAD`, `TOL_GRAD_FAMILY`, `TOL_GRAD_L2`).", "",
         "| case | " + " | ".join(keys) + " | grad maxrel (tensor) | grad l2rel |", "|---|" + "---:|" * (len(keys) + 2)]
    for r in recs:
        o = r["outputs"]
        if not any(k in o for k in keys):
            continue
        gm = max(r["grad_maxrel"].values(), default=[0, ""])
        gl = max(r["grad_l2rel"].values(), default=[0, ""])
        L.append(f"| {r['case']} | " + " | ".join(f"{o[k]:.1e}" if k in o else "" for k in keys)
                 + f" | {gm[0]:.1e} ({gm[1].split('score_model.')[-1]}) | {gl[0]:.1e} |" if gm[0] else
                 f"| {r['case']} | " + " | ".join(f"{o[k]:.1e}" if k in o else "" for k in keys) + " | | |")
    L += ["", "## Worst gradient error per parameter family over all cases", "", "| family | maxrel | tensor | case | l2rel |", "|---|---:|---|---|---:|"]
    for f in fams:
        best = max(((r["grad_maxrel"][f][0], r["grad_maxrel"][f][1], r["case"]) for r in recs if f in r["grad_maxrel"]))
        l2 = max((r["grad_l2rel"][f][0] for r in recs if f in r["grad_l2rel"]), default=0.0)
        L.append(f"| {f} | {best[0]:.1e} | {best[1]} | {best[2]} | {l2:.1e} |")
    L += ["", "## Trajectories and other cases", "", "| case | recorded |", "|---|---|"]
    for r in recs:
        if not any(k in r["outputs"] for k in keys):
            L.append(f"| {r['case']} | {r['outputs']} |")
    open(out, "w").write("\n".join(L) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
