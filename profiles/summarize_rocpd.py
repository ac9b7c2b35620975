#!/usr/bin/env python3
"""Per-kernel statistics (calls, total, average, share) from a rocprofv3 rocpd SQLite database.

rocprofv3 7.2 writes `<name>_results.db` by default; this reproduces the `--stats` kernel table
from it so the summary can be committed as text.  Usage: summarize_rocpd.py results.db > summary.csv
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = con.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, t, a, mn, mx in rows:
        print('"%s",%d,%d,%.1f,%.3f,%d,%d' % (n.replace('"', "'"), c, t, a, 100.0 * t / total, mn, mx))


if __name__ == '__main__':
    main(sys.argv[1])
