#!/bin/bash
# Round 6, VERDICT r05 item 1: what bounds e0_ck on the round-5 sources.  Timing-only builds of the library (results wrong by
# construction; tools/variants/ck_stamps.patch, one -D flag each) against the unmodified build of the same patched sources,
# one process per build, bench.py's event-timed pair and replayed-graph rate; then the in-kernel stamps.
#   for v in base nobwdrows nofwdrows norows nopart unigather noatomic floor; do tools/variants/build_variant.sh ck_stamps exp_$v -DPOVAR_CK_NO_STAMPS <flags>; done
#   tools/variants/build_variant.sh ck_stamps stamps
out=gpurun_out/${1:-r06_evidence}; mkdir -p $out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/bench_driver_flags.json 2> $out/bench_driver_flags.err
for so in build/libpovar_hip_exp_*.so; do
  name=$(basename $so .so); name=${name#libpovar_hip_exp_}
  POVAR_LIB=$so python bench.py --steps 100 --repeats 3 --no-cpu-baseline --no-secondary > $out/exp_$name.json 2> $out/exp_$name.err
  python - "$out/exp_$name.json" "$name" <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:12s} {d['value']:9.1f} terms/s (min {d['value_min']:.0f} max {d['value_max']:.0f})   graph {d['graph_us_per_term']:.2f} us/term   "
      f"e0 pair (events) {1e3 * d['kernel_ms']['e0']:.2f} us   e0 kernel {d['config']['e0_layout']['e0_kernel']} tuned {d['config']['e0_layout']['e0_tune_us']}")
PY
done
POVAR_LIB=build/libpovar_hip_stamps.so python tools/ck_stamps.py venice-1778 > $out/stamps.txt 2> $out/stamps.err
tail -30 $out/stamps.txt
