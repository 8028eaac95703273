"""rocprofv3 --kernel-trace --stats kernel_stats.csv -> the per-kernel markdown table committed under profiles/.
   python tools/kernel_stats_md.py <p_kernel_stats.csv> "<title>" """
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in microseconds; total kernel time {tot / 1e6:.2f} ms over "
          f"{sum(int(r['Calls']) for r in rows)} dispatches.\n")
    print("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|")
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("|", "/")
        if len(name) > 110:
            name = name[:107] + "..."
        print(f"| `{name}` | {int(r['Calls'])} | {float(r['TotalDurationNs']) / 1e3:.1f} | {float(r['AverageNs']) / 1e3:.2f} | "
              f"{float(r['Percentage']):.2f} |")


if __name__ == "__main__":
    main()
