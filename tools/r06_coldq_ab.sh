#!/bin/bash
# Round 6: cold observations of e0_ck through the parent layout's cold view (32-byte q per observation) against a 96-byte partial
# record per cold chunk (POVAR_CK_COLD_RECORDS=1), per graph family.  tools/r06_coldq_ab.sh [tag]
out=gpurun_out/${1:-r06_coldq}; mkdir -p $out; rm -f $out/summary.txt
run() { label=$1; shift; env "$@" > $out/$label.json 2> $out/$label.err
  python3 - "$out/$label.json" "$label" <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
l = d['config']['e0_layout']; c = l['camera_chunks']
print(f"{sys.argv[2]:24s} {d['value']:9.1f} terms/s   graph {d['graph_us_per_term']:.2f} us/term   pair (events) {1e3 * d['kernel_ms']['e0']:.2f} us   kernel {l['e0_kernel']}   records {c['partial_records']}   model {d['roofline']['model_bytes_per_launch'] / 1e6:.0f} MB")
PY
}
B="python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --repeats 3"
for pop in zipf1 zipf0.5 uniform local; do
  run ${pop}_records POVAR_CK_COLD_RECORDS=1 POVAR_E0_CK=1 $B --popularity $pop
  run ${pop}_cold_q POVAR_E0_CK=1 $B --popularity $pop
done
run huber_records POVAR_CK_COLD_RECORDS=1 POVAR_E0_CK=1 $B --robust-norm HUBER
run huber_cold_q POVAR_E0_CK=1 $B --robust-norm HUBER
run final_ck_records POVAR_CK_COLD_RECORDS=1 POVAR_E0_CK=1 $B --problem final-13682 --robust-norm HUBER --huber 20 --steps 5 --warmup 1
run final_ck_cold_q POVAR_E0_CK=1 $B --problem final-13682 --robust-norm HUBER --huber 20 --steps 5 --warmup 1
run final_auto $B --problem final-13682 --robust-norm HUBER --huber 20 --steps 5 --warmup 1
