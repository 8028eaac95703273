"""HBM traffic per kernel launch of the training step, from rocprofv3 --pmc passes, CALIBRATED against known byte counts.

    python tools/pmc_roofline.py <cal_fetch_dir> <cal_write_dir> <step_fetch_dir> <step_write_dir> <out.json>

Inputs are the output directories of four rocprofv3 runs (tools/pmc_roofline.sh): tools/probes/hbm_calib under --pmc FETCH_SIZE
and under --pmc WRITE_SIZE (separate passes: the two counters do not fit one, MI355X_MICROARCH.md "rocprofv3 PMC slots"), and
`bench.py --steps 3` (gradient side stream off, so every dispatch has the chip to itself) under the same two counters.

Calibration (MI355X_MICROARCH.md, HBM): every probe kernel moves exactly 1 GiB each way, so
    factor = 2^30 / (mean counter value x 1024)        (FETCH_SIZE / WRITE_SIZE are reported in KB)
per access pattern: `copy16` = lane-contiguous float4 streams (used for every kernel except the register-chained pair kernels),
`copy16_seg64` = the 16-rows-x-64-bytes-per-instruction pattern of edge_mlp16 / edge_embed (their x loads and h1/h2/y stores).
The JSON holds, per kernel of a step: launches, raw and calibrated bytes fetched / written per launch; bench.py reads it for
roofline.traffic (newest profiles/r*_pmc_traffic.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

GIB = float(1 << 30)
SEG64 = ("edge_mlp16_kernel", "edge_embed_kernel", "edge_embed_bwd_kernel")


def load(path, counter):
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)[0]
    csv.field_size_limit(1 << 30)
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"]) * 1024.0))
    rows.sort()
    return rows


def calib(path, counter):
    by = defaultdict(list)
    for _, name, v in load(path, counter):
        for k in ("copy16_seg64", "copy16", "read16"):
            if k + "(" in name:
                by[k].append(v)
                break
    return {k: {"raw_bytes_mean": sum(v) / len(v), "factor": GIB / (sum(v) / len(v)) if sum(v) else None, "launches": len(v)}
            for k, v in by.items()}


def per_step(rows):
    idx = [i for i, r in enumerate(rows) if "adam_step_kernel" in r[1]]
    assert len(idx) >= 2, "need at least two optimiser steps in the trace"
    by, cnt = defaultdict(float), defaultdict(int)
    for a, b in zip(idx[:-1], idx[1:]):
        for r in rows[a + 1:b + 1]:
            by[r[1]] += r[2]
            cnt[r[1]] += 1
    n = len(idx) - 1
    return {k: (v / n, cnt[k] / n) for k, v in by.items()}, n


def short(name):
    """kernel name without return type, anonymous namespace and argument list (template arguments kept)"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name[:80]


def main():
    cf, cw, sf, sw, out = sys.argv[1:6]
    cal = {"FETCH_SIZE": calib(cf, "FETCH_SIZE"), "WRITE_SIZE": calib(cw, "WRITE_SIZE")}
    fetch, nf = per_step(load(sf, "FETCH_SIZE"))
    write, nw = per_step(load(sw, "WRITE_SIZE"))

    def factor(counter, kernel):
        pat = "copy16_seg64" if any(s in kernel for s in SEG64) else "copy16"
        c = cal[counter].get(pat) or {}
        return (c.get("factor") or 1.0), pat
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        fb, fl = fetch.get(name, (0.0, 0.0))
        wb, wl = write.get(name, (0.0, 0.0))
        launches = max(fl, wl)
        ff, pat = factor("FETCH_SIZE", name)
        wf, _ = factor("WRITE_SIZE", name)
        k = kernels.setdefault(short(name), {"launches_per_step": 0.0, "fetch_raw_per_step": 0.0, "write_raw_per_step": 0.0,
                                             "fetch_per_step": 0.0, "write_per_step": 0.0, "pattern": pat})
        k["launches_per_step"] += launches
        k["fetch_raw_per_step"] += fb
        k["write_raw_per_step"] += wb
        k["fetch_per_step"] += fb * ff
        k["write_per_step"] += wb * wf
    for k in kernels.values():
        n = max(k["launches_per_step"], 1e-9)
        k["bytes_per_launch"] = (k["fetch_per_step"] + k["write_per_step"]) / n
        k["raw_bytes_per_launch"] = (k["fetch_raw_per_step"] + k["write_raw_per_step"]) / n
    tot = {"fetch_raw": sum(k["fetch_raw_per_step"] for k in kernels.values()),
           "write_raw": sum(k["write_raw_per_step"] for k in kernels.values()),
           "fetch": sum(k["fetch_per_step"] for k in kernels.values()),
           "write": sum(k["write_per_step"] for k in kernels.values())}
    res = {"what": "HBM bytes per kernel of one training step (B=30 x N=128), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                   "passes, calibrated on tools/probes/hbm_calib (1 GiB known transfers per access pattern)",
           "steps_averaged": {"fetch": nf, "write": nw}, "calibration": cal, "step_total_bytes": tot, "kernels": kernels}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("calibration:", json.dumps(cal, indent=1))
    print(f"step total: fetched {tot['fetch'] / 1e9:.2f} GB ({tot['fetch_raw'] / 1e9:.2f} raw), written {tot['write'] / 1e9:.2f} GB "
          f"({tot['write_raw'] / 1e9:.2f} raw)")
    for name, k in sorted(kernels.items(), key=lambda kv: -(kv[1]["fetch_per_step"] + kv[1]["write_per_step"]))[:25]:
        print(f"{(k['fetch_per_step'] + k['write_per_step']) / 1e9:8.3f} GB/step  {k['launches_per_step']:6.1f} launches  "
              f"{k['bytes_per_launch'] / 1e6:9.1f} MB/launch  [{k['pattern']}]  {name}")


if __name__ == "__main__":
    main()
