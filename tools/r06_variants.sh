#!/bin/bash
# Round 6: the shipping library and every build/libpovar_hip_exp_*.so (builds of the same sources with one -D constant changed)
# on the headline bench, one process each.  usage: tools/r06_variants.sh <tag> [bench flags...]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for so in povar_amd/libpovar_hip.so build/libpovar_hip_exp_*.so; do
  [ -f $so ] || continue
  name=$(basename $so .so); name=${name#libpovar_hip_exp_}; [ $name = libpovar_hip ] && name=shipping
  POVAR_LIB=$so python bench.py --steps 100 --repeats 3 --no-cpu-baseline --no-secondary "$@" > $out/$name.json 2> $out/$name.err || tail -3 $out/$name.err
  python - "$out/$name.json" "$name" <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:12s} {d['value']:9.1f} terms/s (min {d['value_min']:.0f} max {d['value_max']:.0f})   graph {d['graph_us_per_term']:.2f} us/term   "
      f"e0 pair (events) {1e3 * d['kernel_ms']['e0']:.2f} us   e0 kernel {d['config']['e0_layout']['e0_kernel']} tuned {d['config']['e0_layout']['e0_tune_us']}")
PY
done
