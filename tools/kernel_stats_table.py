#!/usr/bin/env python3
"""Compact table (calls, average us, total ms) from a rocprofv3 *_kernel_stats.csv.  usage: kernel_stats_table.py file.csv [n]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[:n]:
    name = re.sub(r"\(.*", "", r["Name"])
    print(f'{name:60s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:10.1f} us {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
