#!/bin/bash
# e0_ck against e0_lpl over the synthetic graph families (one process per workload: the first two term-loop graphs of a
# process are the ones timed):  tools/ck_sweep.sh <out file> [variant]
out=$1; v=${2:-1}
mkdir -p $(dirname $out); : > $out
run() { POVAR_E0_CK=$v timeout 900 python tools/ck_probe.py "$@" --variants $v >> $out 2>> $out.err < /dev/null; }
run trafalgar-257
run venice-1778
run venice-1778 --robust HUBER
run venice-1778 --popularity local
run venice-1778 --popularity zipf0.5
run venice-1778 --popularity uniform
run final-13682 --robust HUBER --reps 5
run final-13682 --robust HUBER --popularity local --reps 5
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    ck = [v for k, v in d.items() if k.startswith("ck") and isinstance(v, dict)]
    print(f"{d['problem']:14s} {d['robust']:6s} lpl {d['lpl_us_per_term']:8.2f} us  ck {ck[0]['us_per_term'] if ck else float('nan'):8.2f} us  "
          f"chunks {d['ck_chunks']} cold {d['ck_cold_chunks']} nb {d['ck_batches']} e0_rel {ck[0]['e0_rel'] if ck else 0:.1e} inc_rel {ck[0]['inc_rel'] if ck else 0:.1e}")
PY
