#!/bin/bash
# Round 6: which tile of a round each of the sixteen wavefronts walks (an experiment build that takes the permutation from
# POVAR_CK_WPERM: tools/variants -- build/libpovar_hip_wperm.so), e0_ck on venice, graph us per term, one box
out=gpurun_out/${1:-r06_wperm}; mkdir -p $out; rm -f $out/summary.txt
export POVAR_LIB=$PWD/build/libpovar_hip_wperm.so POVAR_E0_CK=1
run() { echo -n "$1 [$2]: " | tee -a $out/summary.txt; POVAR_CK_WPERM=$2 python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['graph_us_per_term'],2))" | tee -a $out/summary.txt; }
run "P0 shipped: groups of four in alternating direction  " 0,1,2,3,7,6,5,4,8,9,10,11,15,14,13,12
run "P1 wavefront w walks tile w (before round 6)         " 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
run "P2 tiles 4-7 first, 11-8, 12-15, 3-0                 " 4,5,6,7,11,10,9,8,12,13,14,15,3,2,1,0
run "P3 tiles 4-7, 8-11, 12-15, 0-3                       " 4,5,6,7,8,9,10,11,12,13,14,15,0,1,2,3
run "P4 tiles 8-11, 4-7, 12-15, 0-3                       " 8,9,10,11,4,5,6,7,12,13,14,15,0,1,2,3
run "P5 tiles 4,8,5,9, 6,10,7,11, 0-3, 12-15              " 4,8,5,9,6,10,7,11,0,1,2,3,12,13,14,15
run "P6 one tile of every quartile per four wavefronts    " 0,4,8,12,1,5,9,13,2,6,10,14,3,7,11,15
run "P7 the same, the expensive one first                 " 4,0,8,12,5,1,9,13,6,2,10,14,7,3,11,15
run "P8 0-3, 4-7, 11-8, 15-12                             " 0,1,2,3,4,5,6,7,11,10,9,8,15,14,13,12
run "P9 0-3, 7-4, 11-8, 12-15                             " 0,1,2,3,7,6,5,4,11,10,9,8,12,13,14,15
run "P0 again                                             " 0,1,2,3,7,6,5,4,8,9,10,11,15,14,13,12
