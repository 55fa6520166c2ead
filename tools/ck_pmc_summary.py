#!/usr/bin/env python3
"""Summary of a tools/ck_pmc.sh directory: per E0 kernel the average duration (kernel trace), HBM bytes per launch
(2 * FETCH_SIZE + WRITE_SIZE, counter unit KB: MI355X_MICROARCH.md) and the SQ counters."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, pat):
    fs = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def pmc(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    d = sys.argv[1]
    keep = lambda k: "e0_" in k or "cam_cold" in k
    out = {}
    ks = find(os.path.join(d, "kt"), "*kernel_stats.csv")
    if ks:
        for r in csv.DictReader(open(ks)):
            n = r["Name"].split("(")[0]
            if keep(n):
                out.setdefault(n, {})["avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
                out[n]["calls"] = int(r["Calls"])
    f, w = find(os.path.join(d, "fetch"), "*counter_collection.csv"), find(os.path.join(d, "write"), "*counter_collection.csv")
    if f and w:
        pf, pw = pmc(f), pmc(w)
        for k in pf:
            if keep(k):
                fe, wr = pf[k]["FETCH_SIZE"] * 1024, pw.get(k, {}).get("WRITE_SIZE", 0) * 1024
                out.setdefault(k, {}).update({"fetch2_MB": round(2 * fe / 1e6, 1), "write_MB": round(wr / 1e6, 1), "hbm_MB": round((2 * fe + wr) / 1e6, 1)})
    sq = find(os.path.join(d, "sq"), "*counter_collection.csv")
    if sq:
        for k, v in pmc(sq).items():
            if keep(k):
                out.setdefault(k, {})["sq"] = {c.replace("SQ_", ""): round(x / 1e6, 2) for c, x in v.items()}
    for k, v in out.items():
        if "avg_us" in v and "hbm_MB" in v:
            v["TBps"] = round(v["hbm_MB"] / v["avg_us"] / 1e3, 3)
    json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
    for k in sorted(out):
        print(k, json.dumps(out[k]))


if __name__ == "__main__":
    main()
