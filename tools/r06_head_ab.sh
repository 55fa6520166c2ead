#!/bin/bash
out=gpurun_out/${1:-r06_head_ab}; mkdir -p $out; rm -f $out/summary.txt
run() { label=$1; shift; env "$@" python bench.py --steps 100 --repeats 3 --no-cpu-baseline --no-secondary > $out/$label.json 2> $out/$label.err
  python - "$out/$label.json" "$label" <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
l = d['config']['e0_layout']
print(f"{sys.argv[2]:14s} {d['value']:9.1f} terms/s   graph {d['graph_us_per_term']:.2f} us/term   e0 (events) {1e3 * d['kernel_ms']['e0']:.2f} us  binv {1e3 * d['kernel_ms']['binv_axpy']:.2f}")
PY
}
run prev POVAR_LIB=build/libpovar_hip_exp_prev.so
run nohead POVAR_LIB=build/libpovar_hip_exp_nohead.so POVAR_CK_HEAD=0
run head0 POVAR_CK_HEAD=0
run head1 POVAR_CK_HEAD=1
run auto POVAR_X=1
run prev2 POVAR_LIB=build/libpovar_hip_exp_prev.so
