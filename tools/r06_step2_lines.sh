#!/bin/bash
# step-2 bench lines with the library's stride and with 1536 forced (same box)  -> gpurun_out/<tag>/
out=gpurun_out/${1:-r06_step2w}; mkdir -p $out
B="python3 bench.py --no-secondary --no-cpu-baseline --step 2"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['e0_layout']; print(round(d['value']), 'terms/s', round(d['graph_us_per_term'],2), 'us per term;', c.get('camera_chunks_step2'), 'kernel', c.get('e0_kernel_step2'), c.get('e0_tune_us_step2'))"; }
run() { label=$1; shift; echo -n "$label: " | tee -a $out/summary.txt; env "$@" $B 2> $out/err.txt | tee $out/last.json | line | tee -a $out/summary.txt; }
run "venice step 2, library's choice          " A=1
run "venice step 2, POVAR_CKH_STRIDE=1536     " POVAR_CKH_STRIDE=1536
B="$B --robust-norm HUBER"
run "venice step 2 HUBER, library's choice    " A=1
run "venice step 2 HUBER, POVAR_CKH_STRIDE=1536" POVAR_CKH_STRIDE=1536
B="python3 bench.py --no-secondary --no-cpu-baseline --step 2 --popularity local"
run "venice local step 2, library's choice    " A=1
run "venice local step 2, 1536                " POVAR_CKH_STRIDE=1536
B="python3 bench.py --no-secondary --no-cpu-baseline --step 2 --popularity uniform"
run "venice uniform step 2, library's choice  " A=1
run "venice uniform step 2, 1536              " POVAR_CKH_STRIDE=1536
