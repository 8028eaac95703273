"""Turn a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) into the per-kernel
summary committed under profiles/.   python tools/prof_summary.py <results.db> <out.md> [title]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in microseconds; total kernel time "
                f"{tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} dispatches.\n\n")
        f.write("| kernel | calls | total us | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, total, avg, pct in rows:
            name = name.replace("(anonymous namespace)::", "").replace("|", "/")
            if len(name) > 110:
                name = name[:107] + "..."
            f.write(f"| `{name}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |\n")


if __name__ == "__main__":
    main()
