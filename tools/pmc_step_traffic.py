"""Whole-step HBM traffic from rocprofv3 --pmc passes (FETCH_SIZE in one pass, WRITE_SIZE in another):
sum over every dispatch between two adam_step_kernel launches, averaged over the steps in the trace.
   python tools/pmc_step_traffic.py <fetch_dir> <write_dir> [pairs_per_step]
FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streaming reads by 2x on
gfx950 -- both the raw and the x2 figure are printed."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(path, counter):
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)[0]
    csv.field_size_limit(1 << 30)
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def per_step(rows):
    idx = [i for i, r in enumerate(rows) if "adam_step_kernel" in r[1]]
    if len(idx) < 2:
        return sum(r[2] for r in rows), 1, {}
    tot, by = 0.0, defaultdict(float)
    for a, b in zip(idx[:-1], idx[1:]):
        for r in rows[a + 1:b + 1]:
            tot += r[2]
            by[r[1][:70]] += r[2]
    n = len(idx) - 1
    return tot / n, n, {k: v / n for k, v in by.items()}


def main():
    fetch, nf, byf = per_step(load(sys.argv[1], "FETCH_SIZE"))
    write, nw, byw = per_step(load(sys.argv[2], "WRITE_SIZE"))
    pairs = float(sys.argv[3]) if len(sys.argv) > 3 else 30 * 128 * 128
    kb = 1024.0
    print(f"steps averaged: fetch {nf}, write {nw}")
    print(f"FETCH_SIZE per step: {fetch * kb / 1e9:.3f} GB raw, {2 * fetch * kb / 1e9:.3f} GB with the gfx950 x2 correction")
    print(f"WRITE_SIZE per step: {write * kb / 1e9:.3f} GB")
    for corr, name in ((1, "raw"), (2, "x2")):
        tot = (corr * fetch + write) * kb
        print(f"HBM bytes per step ({name}): {tot / 1e9:.3f} GB = {tot / pairs:.0f} B/pair; algorithmic 3 x 5632 B/pair = "
              f"{3 * 5632 * pairs / 1e9:.3f} GB -> ratio {tot / (3 * 5632 * pairs):.2f}")
    print("top kernels by FETCH_SIZE (KB/step):")
    for k, v in sorted(byf.items(), key=lambda kv: -kv[1])[:12]:
        print(f"   {v:12.0f}  {k}")
    print("top kernels by WRITE_SIZE (KB/step):")
    for k, v in sorted(byw.items(), key=lambda kv: -kv[1])[:12]:
        print(f"   {v:12.0f}  {k}")


if __name__ == "__main__":
    main()
