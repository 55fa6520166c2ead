#!/bin/bash
# Round 6, VERDICT r05 item 2: the compute part of one rank's term on the venice shard of 1/8 (and 1/4), per kernel choice.
out=gpurun_out/${1:-r06_shard}; mkdir -p $out
export TMPDIR=/tmp
run() {  # label, env...
  label=$1; shift
  for n in 8 4; do
    echo -n "$label world=$n: " | tee -a $out/summary.txt
    env POVAR_FORCE_COMM=1 POVAR_GRAPH_COMM=1 "$@" python3 tools/shard_term_time.py $n venice-1778 2>&1 | grep "world=" | sed 's/.*obs; //' | tee -a $out/summary.txt
  done
}
run "default            "
run "e0_ck<16,2> forced " POVAR_E0_CK=1
run "e0_ck<12,2,DB>     " POVAR_E0_CK=3
run "e0_ck<8,2,DB>      " POVAR_E0_CK=6
run "e0_lpl forced      " POVAR_E0_CK=0
run "128 workgroups, ck1" POVAR_E0_CK=1 POVAR_E0_WGS=128
run "64 workgroups, ck1 " POVAR_E0_CK=1 POVAR_E0_WGS=64
run "128 workgroups, ck6" POVAR_E0_CK=6 POVAR_E0_WGS=128
POVAR_FORCE_COMM=1 POVAR_GRAPH_COMM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/shard_term_time.py 8 venice-1778 > /dev/null 2> $out/kt.err
f=$(ls $out/kt/*/*kernel_stats.csv | head -1); python3 tools/kernel_stats_table.py $f 12 | tee -a $out/summary.txt
