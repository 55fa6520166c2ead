#!/bin/bash
# A/B of an experiment build (build/libpovar_hip_<variant>.so) against the shipped library on the headline workload, same box:
#   tools/r06_variant_ab.sh <tag> <variant> ["bench args" ...]
out=gpurun_out/$1; mkdir -p $out; rm -f $out/summary.txt
V=POVAR_LIB=$PWD/build/libpovar_hip_$2.so; shift 2
run() { echo -n "$* : " | tee -a $out/summary.txt; env "$@" python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --repeats 3 $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['graph_us_per_term'],2))" | tee -a $out/summary.txt; }
[ $# -eq 0 ] && set -- ""
for ARGS in "$@"; do
  echo "== bench.py $ARGS" | tee -a $out/summary.txt
  run POVAR_E0_CK=1
  run POVAR_E0_CK=1 $V
  run POVAR_E0_CK=1
  run POVAR_E0_CK=1 $V
done
