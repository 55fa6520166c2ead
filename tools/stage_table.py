#!/usr/bin/env python3
"""Per-kernel table of a tools/stage_rooflines.sh directory: calls, average duration (rocprofv3 --kernel-trace --stats),
HBM bytes per launch (2 * FETCH_SIZE + WRITE_SIZE, counter unit KB, MI355X_MICROARCH.md's gfx950 correction), GB/s and the
fraction of the 8 TB/s peak.  Writes <dir>/stages.json and prints a markdown table."""
import collections
import csv
import glob
import json
import os
import sys

PEAK = 8000.0


def find(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None  # the newest run (gpurun merges runs of one tag)


def pmc(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}


def main():
    d, args = sys.argv[1], sys.argv[2:]
    ks = list(csv.DictReader(open(find(os.path.join(d, "kt"), "*kernel_stats.csv"))))
    fetch = pmc(find(os.path.join(d, "fetch"), "*counter_collection.csv"), "FETCH_SIZE")
    write = pmc(find(os.path.join(d, "write"), "*counter_collection.csv"), "WRITE_SIZE")
    rows = []
    for r in ks:
        name = r["Name"].split("(")[0]
        us = float(r["AverageNs"]) / 1e3
        if us < 3.0 and "cam_" not in name:
            continue
        b = 2 * fetch.get(name, 0.0) + write.get(name, 0.0)
        rows.append({"kernel": name.replace("void ", "").replace("povar::", ""), "calls": int(r["Calls"]), "avg_us": us,
                     "fetch_raw_bytes": fetch.get(name, 0.0), "write_bytes": write.get(name, 0.0), "hbm_bytes": b,
                     "GBps": b / us / 1e3 if us > 0 else 0.0, "frac": b / us / 1e3 / PEAK if us > 0 else 0.0,
                     "pct_of_gpu_time": float(r["Percentage"])})
    json.dump({"command": "tools/stage_loop.py " + " ".join(args), "peak_GBps": PEAK, "kernels": rows},
              open(os.path.join(d, "stages.json"), "w"), indent=1)
    print(f"| kernel | calls | avg µs | HBM MB / launch (2·FETCH + WRITE) | GB/s | fraction of 8 TB/s | % of GPU time |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| `{r['kernel']}` | {r['calls']} | {r['avg_us']:.1f} | {r['hbm_bytes'] / 1e6:.1f} | {r['GBps']:.0f} | {r['frac']:.2f} | {r['pct_of_gpu_time']:.1f} |")


if __name__ == "__main__":
    main()
