#!/usr/bin/env python3
"""Table of tools/micro/fetch_calibration.hip: known bytes per launch against FETCH_SIZE / WRITE_SIZE (counter unit KB).

    fetch_calibration_table.py <fetch pass dir> <write pass dir>

For each kernel: bytes the launch moves by construction, the counter, and counter / bytes -- the factor
tools/pmc_to_traffic.py has to divide by for that access pattern (1.0 = exact, 0.5 = the counter sees half)."""
import collections
import csv
import glob
import json
import os
import sys

BIG, SMALL, NG = 1 << 30, 96 << 20, 8 << 20
KNOWN = {  # kernel name fragment -> (read bytes, written bytes) per launch
    "read_big<float>": (BIG, 0), "read_small<float>": (SMALL, 0),
    "read_big<double>": (BIG, 0), "read_small<double>": (SMALL, 0),
    "read_big<HIP_vector_type<double, 2": (BIG, 0), "read_small<HIP_vector_type<double, 2": (SMALL, 0),
    "write_big<double>": (0, BIG), "write_small<double>": (0, SMALL),
    "write_big<HIP_vector_type<double, 2": (0, BIG), "write_small<HIP_vector_type<double, 2": (0, SMALL),
    "gather32_big": (NG * 36, 0), "gather32_small": (NG * 36, 0), "gather168_small": (NG // 8 * 172, 0),
    "scatter32_big": (NG * 4, NG * 32),
}


def pmc(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch, write = pmc(sys.argv[1]), pmc(sys.argv[2])
    out = {}
    print(f"{'kernel':44s} {'read MB':>9s} {'FETCH MB':>9s} {'ratio':>6s} {'write MB':>9s} {'WRITE MB':>9s} {'ratio':>6s}")
    for frag, (rb, wb) in KNOWN.items():
        name = next((k for k in fetch if frag in k), None)
        if name is None:
            continue
        f, w = fetch[name], write.get(name, 0.0)
        out[frag] = {"read_bytes": rb, "fetch_size_bytes": f, "fetch_ratio": f / rb if rb else None,
                     "write_bytes": wb, "write_size_bytes": w, "write_ratio": w / wb if wb else None}
        print(f"{frag[:44]:44s} {rb / 1e6:9.1f} {f / 1e6:9.1f} {f / rb if rb else float('nan'):6.3f} {wb / 1e6:9.1f} {w / 1e6:9.1f} "
              f"{w / wb if wb else float('nan'):6.3f}")
    json.dump(out, sys.stdout if len(sys.argv) < 4 else open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
