#!/bin/bash
# Round 6, VERDICT r05 item 2 ("size the grid to the shard"): where between 1/8 and 1/4 of venice do 128 workgroups stop paying,
# and what do they do to the small problems?  -> gpurun_out/<tag>/summary.txt
out=gpurun_out/${1:-r06_wgs}; mkdir -p $out
export TMPDIR=/tmp
run() {  # label, worlds, env...
  label=$1; worlds=$2; shift 2
  for n in $worlds; do
    echo -n "$label world=$n: " | tee -a $out/summary.txt
    env POVAR_FORCE_COMM=1 POVAR_GRAPH_COMM=1 "$@" python3 tools/shard_term_time.py $n venice-1778 2>&1 | grep "world=" | sed 's/.*landmarks \/ //' | tee -a $out/summary.txt
  done
}
run "256 workgroups" "16 8 6 5 4"
run "192 workgroups" "16 8 6 5 4" POVAR_E0_WGS=192
run "128 workgroups" "16 8 6 5 4" POVAR_E0_WGS=128
run "96 workgroups " "16 8" POVAR_E0_WGS=96
B="python3 bench.py --no-secondary --no-cpu-baseline --repeats 3"
for p in trafalgar-257 ladybug-49; do
  for w in 256 128; do
    for res in 1 0; do
      echo -n "$p workgroups=$w POVAR_RES=$res: " | tee -a $out/summary.txt
      POVAR_E0_WGS=$w POVAR_RES=$res $B --problem $p 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'terms/s', d.get('graph_us_per_term'))" | tee -a $out/summary.txt
    done
  done
done
