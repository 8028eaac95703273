"""Summarise a rocprofv3 --pmc csv (pmc_counter_collection.csv): mean counter value per launch for kernels whose
name contains a substring.   python tools/pmc_summary.py <dir-or-csv> <kernel-substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path, sub = sys.argv[1], sys.argv[2]
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)[0]
    csv.field_size_limit(1 << 30)
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            if sub in row["Kernel_Name"]:
                key = (row["Kernel_Name"][:90], row["Grid_Size"], row["Workgroup_Size"])
                acc[key][row["Counter_Name"]].append((float(row["Counter_Value"]),
                                                      int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    for key, ctr in acc.items():
        print(key)
        for name, vals in sorted(ctr.items()):
            v = [x[0] for x in vals]
            t = [x[1] for x in vals]
            print(f"   {name:32s} n={len(v):3d} mean={sum(v) / len(v):16.1f}  dur_us={sum(t) / len(t) / 1e3:10.1f}")


if __name__ == "__main__":
    main()
