#!/bin/bash
# What FETCH_SIZE is made of, on the step-1 and step-2 term loops: the L2's request / hit / miss counters and its
# memory-side read requests by size, one pass each (the TCC block has four counters):  tools/pmc_tcc_probe.sh <out dir>
out=${1:-gpurun_out/r04tcc}
cd "$(dirname "$0")/.." || exit 1
mkdir -p $out
export TMPDIR=/tmp POVAR_NO_GRAPH=1 POVAR_E0_CK=0
timeout 120 rocprofv3 --list-avail > $out/avail.txt 2>&1 < /dev/null
grep -o "TCC_[A-Z0-9_a-z]*" $out/avail.txt | sort -u > $out/tcc_counters.txt
pass() {  # name, counters..., then -- bench args
  local name=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  timeout 600 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d $out/$name -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 "$@" > /dev/null 2> $out/$name.err < /dev/null
}
for s in 1 2; do
  pass s${s}_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- --step $s
  pass s${s}_hit TCC_HIT_sum TCC_MISS_sum -- --step $s
  pass s${s}_req TCC_REQ_sum TCC_READ_sum -- --step $s
  pass s${s}_fetch FETCH_SIZE -- --step $s
done
python3 - $out <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for s in (1, 2):
    row = {}
    for p in ("ea", "hit", "req", "fetch"):
        for f in glob.glob(f"{out}/s{s}_{p}/*/*counter_collection.csv"):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "e0_lpl" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                row[k] = sum(v) / len(v)
    print(f"step {s}:", {k: round(v) for k, v in row.items()})
PY
