"""Build-container measurement (needs /root/reference): the CPU port (oracle/framediff_oracle.py) against the UNMODIFIED reference
on the same host cores and the same workload as bench.py's cpu_baseline (fwd + DSM loss + bwd, B=4 x N=128, 4 blocks), median of 5
steps each at the same thread count.  Writes profiles/r04_cpu_port_vs_reference.json, which bench.py copies into
cpu_baseline.port_vs_reference so that a `kind: "port"` line measured on the GPU box documents how the port relates to the real
reference.   python tools/cpu_port_vs_reference.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    ncpu = os.cpu_count()
    res = {}
    for kind in ("reference", "port"):
        r = bench.cpu_baseline(128, 4, 4, budget_s=120.0, reps=5, force_kind=("port" if kind == "port" else None))
        assert r["kind"] == kind, r
        res[kind] = {"residues_per_s": r["value"], "threads": r["cores"], "sample": r["sample"]}
    out = {"port_over_reference": round(res["port"]["residues_per_s"] / res["reference"]["residues_per_s"], 3),
           "reference_residues_per_s": res["reference"]["residues_per_s"], "port_residues_per_s": res["port"]["residues_per_s"],
           "threads": (res["reference"]["threads"], res["port"]["threads"]), "logical_cpus": ncpu, "torch": torch.__version__,
           "where": "build container (the reference checkout exists only here)", "workload": res["reference"]["sample"]}
    with open(os.path.join(ROOT, "profiles", "r04_cpu_port_vs_reference.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
