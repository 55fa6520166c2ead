#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd database (`*_results.db`),
the same table `--stats` prints as kernel_stats.csv.   usage: rocpd_kernel_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = con.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for name, calls, tot, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)
        lines.append(f'"{short}",{calls},{tot},{avg:.1f},{mn},{mx},{100.0 * tot / total:.2f}')
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
