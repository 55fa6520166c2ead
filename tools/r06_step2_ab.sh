#!/bin/bash
out=gpurun_out/${1:-r06_step2_ab}; mkdir -p $out; rm -f $out/summary.txt
run() { label=$1; shift; env "$@" python bench.py --step 2 --steps 100 --repeats 3 --no-cpu-baseline --no-secondary > $out/$label.json 2> $out/$label.err
  python - "$out/$label.json" "$label" <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:14s} {d['value']:9.1f} terms/s   graph {d['graph_us_per_term']:.2f} us/term   e0 (events) {1e3 * d['kernel_ms']['e0']:.2f} us  kernel_h {d['config']['e0_layout']['e0_kernel_step2']} tuned {d['config']['e0_layout']['e0_tune_us_step2']}")
PY
}
run prev POVAR_LIB=build/libpovar_hip_exp_prev.so
run new POVAR_X=1
run prev2 POVAR_LIB=build/libpovar_hip_exp_prev.so
run new2 POVAR_X=1
