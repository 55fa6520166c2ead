#!/bin/bash
# Round 6: tiles of a round dealt to the wavefronts so that every SIMD gets a balanced share (snake over the groups of four) against
# the shipped order (wavefront w walks tile w: SIMD 0 hosts the longest tile of every group of four); same box, graph us per term
out=gpurun_out/${1:-r06_snake}; mkdir -p $out; rm -f $out/summary.txt
run() { echo -n "$* : " | tee -a $out/summary.txt; env "$@" python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --repeats 3 $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['graph_us_per_term'],2))" | tee -a $out/summary.txt; }
V=POVAR_LIB=$PWD/build/libpovar_hip_${2:-snake}.so
for ARGS in "" "--robust-norm HUBER" "--step 2" "--popularity local" "--popularity uniform"; do
  echo "== bench.py $ARGS" | tee -a $out/summary.txt
  run POVAR_E0_CK=1
  run POVAR_E0_CK=1 $V
  run POVAR_E0_CK=1
  run POVAR_E0_CK=1 $V
done
