#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 rocpd database: every kernel between two launches of the step's first
kernel (step_begin_kernel; rng_fill_kernel before round 3), in start order, with its duration, the idle gap before it on its own queue, and how much of
the step's wall time had at least one kernel running.  Usage: timeline.py results.db [step index from the end = 2]"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    rows = db.execute(f"select {name}, start, end, {qcol or 0} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "step_begin_kernel" in r[0]]
    if len(marks) < 3:      # a build without the fused step head
        marks = [i for i, r in enumerate(rows) if "rng_fill" in r[0] or "tick_kernel" in r[0]]
    lo, hi = marks[-back - 1], marks[-back]
    step = rows[lo:hi]
    t0, t1 = step[0][1], rows[hi][1]
    print(f"columns: {cols}")
    print(f"step of {len(step)} kernels, wall {1e-3 * (t1 - t0):.1f} us, kernel time {1e-3 * sum(r[2] - r[1] for r in step):.1f} us")
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    for _, s, e, _ in step:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"some kernel running for {1e-3 * busy:.1f} us; idle {1e-3 * (t1 - t0 - busy):.1f} us")
    last_end = {}
    queues = sorted({r[3] for r in step})
    print("queues:", queues)
    fam = {}
    for n, s, e, q in step:
        gap = s - last_end.get(q, s)
        last_end[q] = e
        f = short(n)
        d = fam.setdefault((q, f), [0, 0, 0])
        d[0] += 1; d[1] += e - s; d[2] += max(gap, 0)
        if len(sys.argv) > 3:
            print(f"{1e-3 * (s - t0):9.1f} q{queues.index(q)} {1e-3 * (e - s):7.1f} us gap {1e-3 * gap:6.1f}  {f}")
    print("per queue and kernel: launches, total us, total gap before (us)")
    for (q, f), d in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  q{queues.index(q)} {f:45s} {d[0]:4d} {1e-3 * d[1]:8.1f} {1e-3 * d[2]:8.1f}")
    for q in queues:
        tot = sum(d[1] for (qq, _), d in fam.items() if qq == q)
        gaps = sum(d[2] for (qq, _), d in fam.items() if qq == q)
        print(f"queue q{queues.index(q)}: kernel time {1e-3 * tot:.1f} us, gaps {1e-3 * gaps:.1f} us")


if __name__ == "__main__":
    main()
