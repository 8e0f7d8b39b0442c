#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this prints the per-kernel summary of
`--kernel-trace --stats` as CSV (same columns as rocprofv3's *_kernel_stats.csv).

usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, calls, tot, avg, mn, mx in rows:
        print('"%s",%d,%d,%.3f,%.2f,%d,%d' % (name, calls, tot, avg, 100.0 * tot / total, mn, mx))


if __name__ == "__main__":
    main(sys.argv[1])
