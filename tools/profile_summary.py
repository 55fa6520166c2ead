#!/usr/bin/env python3
"""Summaries of a tools/profile_e0.sh run directory: kernel_stats (top kernels), SQ counters per kernel,
FETCH_SIZE / WRITE_SIZE per kernel (bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE, counter unit KB)."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return fs[0] if fs else None


def pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    d = sys.argv[1]
    out = {}
    ks = find(os.path.join(d, "kt"), "*kernel_stats.csv")
    if ks:
        rows = list(csv.DictReader(open(ks)))
        out["kernel_stats"] = [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:12]]
    sq = find(os.path.join(d, "sq"), "*counter_collection.csv")
    if sq:
        out["sq"] = {k: v for k, v in pmc(sq).items() if "e0_" in k or "cam_cold" in k}
    f, w = find(os.path.join(d, "fetch"), "*counter_collection.csv"), find(os.path.join(d, "write"), "*counter_collection.csv")
    if f and w:
        pf, pw = pmc(f), pmc(w)
        out["hbm_bytes_per_launch"] = {k: {"fetch_raw": pf[k]["FETCH_SIZE"] * 1024, "write": pw.get(k, {}).get("WRITE_SIZE", 0) * 1024,
                                           "hbm": 2 * pf[k]["FETCH_SIZE"] * 1024 + pw.get(k, {}).get("WRITE_SIZE", 0) * 1024}
                                       for k in pf if "e0_" in k or "cam_cold" in k}
    json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
