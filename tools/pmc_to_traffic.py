#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md's HBM section prescribes) into profiles/traffic.json for bench.py.

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (counter unit KB).  Calibrated on this code's own access patterns
(tools/micro/fetch_calibration.hip, profiles/r04_fetch_calibration.txt, MI355X / ROCm 7.2):
  * coalesced reads of 4, 8 AND 16 bytes per lane: FETCH_SIZE = 0.500 x the bytes (1 GiB footprint and a 96 MiB one swept
    repeatedly alike: Infinity-Cache hits are counted) -> the factor 2 is exact for every stream the E0 kernels read
    (uv rows 16 B, slot / cw words 4 B, landmark-record entries 8 B);
  * coalesced stores of 8 and 16 bytes per lane and scattered 32-byte stores: WRITE_SIZE = 1.00 .. 1.03 x the bytes;
  * random 32-byte gathers (cm_gram's landmarks, q of cold observations): FETCH_SIZE = 64 bytes per 32-byte record, i.e.
    2 x FETCH_SIZE = 4 x the bytes asked for -- the 128-byte line each gather drags in, if the counter's halving holds for
    them too; an upper bound otherwise (the kernels concerned are marked in DESIGN.md);
  * gathers from an L2-resident table (e0_ck's camera records, 341 KB): not counted (0.02 x) -- FETCH_SIZE sees L2 misses.

usage: pmc_to_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <key> [out.json]
"""
import collections
import csv
import json
import os
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in agg.items()}


def main():
    fetch, write, key = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic.json")
    e0 = [k for k in fetch if "e0_lpl" in k or "e0_ck" in k or "e0_lm_cached" in k or "OpE0Tiles" in k or "cm_scatter" in k or "cam_cold_sum" in k]
    table = {k: {"fetch_raw_bytes": fetch[k], "write_bytes": write.get(k, 0.0),
                 "hbm_bytes": 2 * fetch[k] + write.get(k, 0.0)} for k in sorted(fetch)}
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    problem, mode, world = key.split(":")[:3]  # further fields (step2, HUBER, local, ...) only qualify the key
    ck = [f for f in key.split(":") if f.startswith("ck")]   # ck<variant> (step 1: e0_ck) or ckh1 (step 2: e0_ck_h)
    lm = [k for k in e0 if ((("e0_ck" in k) if ck else ("e0_lpl" in k or "e0_lm_cached" in k)) if mode != "tiles" else "OpE0Tiles" in k)]
    # camera-major half of E0: cm_scatter (deterministic modes) or cam_cold_sum[_binv] (LDSACC modes; the fused
    # kernel also carries the 2 MB of B^-1 reads of the AXPY)
    cm = [k for k in e0 if ("cam_cold_sum" in k if "ldsacc" in mode else "cm_scatter" in k)]
    if any("cam_cold_sum_binv" in k for k in cm):  # the term loop's fused kernel; plain cam_cold_sum is prepare_Hb's
        cm = [k for k in cm if "cam_cold_sum_binv" in k]
    data[key] = sum(table[k]["hbm_bytes"] for k in lm + cm)
    data.setdefault("_per_kernel", {})[key] = {k: table[k] for k in lm + cm}
    # stamp: bench.py reports the figure only while the kernel sources are the ones it was measured on
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    data.setdefault("_source_sha", {})[key] = bench.kernel_source_sha()
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({key: data[key]}))


if __name__ == "__main__":
    main()
