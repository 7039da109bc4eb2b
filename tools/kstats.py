#!/usr/bin/env python
"""Per-kernel statistics (calls, total / average / min / max duration) from a rocprofv3 rocpd database
(``*_results.db``), written as the same CSV the ``--stats`` CSV output has.  Usage: kstats.py results.db [out.csv]"""
import sqlite3
import sys


def kernel_stats(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      f"from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2], r[3], 100.0 * r[2] / total, r[4], r[5]) for r in rows]


if __name__ == "__main__":
    rows = kernel_stats(sys.argv[1])
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for r in rows:
        out.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % r)
