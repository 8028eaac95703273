"""GPU busy time vs wall time of a training step from a rocprofv3 --kernel-trace csv: steps are delimited by the
adam_step_kernel dispatches.  Reports, per step, the span, the union of the kernel intervals (GPU busy on any stream),
the idle gap time and the number of dispatches; and the distribution of gaps between consecutive kernels.
   python tools/step_gap.py <..._kernel_trace.csv>"""
import csv
import sys


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adam_step_kernel" in r[2]]
    print(f"{len(rows)} dispatches, {len(marks)} optimiser steps")
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        span = seg[-1][1] - rows[a][1]
        busy, cur_s, cur_e = 0, None, None
        gaps = []
        for s, e, _, _q in seg:
            if cur_e is None:
                cur_s, cur_e = s, e
            elif s <= cur_e:
                cur_e = max(cur_e, e)
            else:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
                cur_s, cur_e = s, e
        busy += cur_e - cur_s
        tot = sum(e - s for s, e, _, _q in seg)
        byq = {}
        for s_, e_, n_, q_ in seg:
            byq.setdefault(q_, [0, 0])
            byq[q_][0] += e_ - s_
            byq[q_][1] += 1
        gaps.sort()
        med = gaps[len(gaps) // 2] if gaps else 0
        print(f"step: span {span / 1e6:7.3f} ms  busy(any stream) {busy / 1e6:7.3f} ms  idle {100 * (span - busy) / span:5.1f} %  "
              f"sum of kernel durations {tot / 1e6:7.3f} ms  dispatches {len(seg)}  gaps: n={len(gaps)} median {med / 1e3:.2f} us "
              f"p90 {gaps[int(0.9 * len(gaps))] / 1e3 if gaps else 0:.2f} us  sum {sum(gaps) / 1e6:.3f} ms  per queue (ms, n): "
              + ", ".join(f"{q}: {v[0] / 1e6:.2f}/{v[1]}" for q, v in sorted(byq.items())))
    # the largest kernels of the last side-stream step
    if len(marks) > 4:
        a, b = marks[3], marks[4]
        agg = {}
        for s_, e_, n_, q_ in rows[a + 1:b + 1]:
            k = (q_, n_.replace("(anonymous namespace)::", "")[:70])
            agg.setdefault(k, [0, 0])
            agg[k][0] += e_ - s_
            agg[k][1] += 1
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
            print(f"  q{k[0]}  {v[0] / 1e3:9.1f} us  n={v[1]:3d}  {k[1]}")
    # timeline of that step: every dispatch in start order -- offset from the step start, duration, the idle time before it
    # on its own queue, and the side-queue kernels running at its start
    if len(marks) > 4 and len(sys.argv) > 2:
        a, b = marks[3], marks[4]
        seg = rows[a + 1:b + 1]
        t0 = rows[a][1]
        mainq = max(set(q for _, _, _, q in seg), key=lambda q: sum(1 for r in seg if r[3] == q))
        last_end = {}
        with open(sys.argv[2], "w") as f:
            for s_, e_, n_, q_ in seg:
                gap = s_ - last_end.get(q_, t0)
                last_end[q_] = e_
                co = [n2.replace("(anonymous namespace)::", "")[:24] for s2, e2, n2, q2 in seg if q2 != q_ and s2 <= s_ < e2]
                name = n_.replace("(anonymous namespace)::", "").replace("void ", "")[:78]
                f.write(f"{(s_ - t0) / 1e6:8.3f} ms  {'M' if q_ == mainq else 's'}  {(e_ - s_) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {name}"
                        + (f"   || {', '.join(co)}" if co else "") + "\n")


if __name__ == "__main__":
    main()
