#!/bin/bash
# Per-kernel durations of a whole `bal` run (both LM steps) on a synthetic BAL shape:
#   tools/bal_kernel_trace.sh <tag> <problem> [bal flags...]   -> gpurun_out/<tag>/kt (rocprofv3 --kernel-trace --stats)
set -u
tag=$1; name=$2; shift 2
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 tools/run_bal_config.py $name "$@" > $out/bal_summary.json 2> $out/bal.err   # also writes the data file
# the file run_bal_config.py wrote for THIS problem (same formula: $TMPDIR/problem-<cams>-<landmarks>-pre.txt)
f=$(python3 -c "import sys, os, tempfile; sys.path.insert(0, '.'); from povar_amd import synth; c, l, _ = synth.BAL_SHAPES['$name']; print(os.path.join(tempfile.gettempdir(), f'problem-{c}-{l}-pre.txt'))")
[ -f "$f" ] || { echo "missing $f" >&2; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- bin/bal --input $f --quiet "$@" > $out/bal_trace.stdout 2> $out/kt.err
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/kernel_stats.txt", "w") as o:
    for r in rows[:40]:
        o.write(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):6d} avg_us {float(r["AverageNs"])/1e3:10.1f} pct {float(r["Percentage"]):6.2f}\n')
print(open(out + "/kernel_stats.txt").read())
PY
