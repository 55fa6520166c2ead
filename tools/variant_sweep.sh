#!/bin/bash
# Times build/variants/libpovar_hip_<name>.so (experiment builds of the library, see the -D flags in the commit that
# made them) against the committed library on the headline bench.  usage: tools/variant_sweep.sh <tag> [bench flags...]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cp povar_amd/libpovar_hip.so /tmp/libpovar_hip_base.so
for so in /tmp/libpovar_hip_base.so build/variants/libpovar_hip_*.so; do
  name=$(basename $so .so); name=${name#libpovar_hip_}
  cp $so povar_amd/libpovar_hip.so
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --no-secondary "$@" > $out/$name.$rep.json 2> $out/$name.err
    python - "$out/$name.$rep.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:10s} {d['value']:9.1f} terms/s   e0 pair {1e3 * d['kernel_ms']['e0']:.1f} us")
PY
  done
done
cp /tmp/libpovar_hip_base.so povar_amd/libpovar_hip.so
