#!/bin/bash
# final-13682 step 2: e0_ck_h on the wide stride (9 batches, 314 accumulators) against 1536 (12 batches, 501), forced and the library's choice
out=gpurun_out/${1:-r06_final_s2}; mkdir -p $out; rm -f $out/summary.txt
B="python3 bench.py --no-secondary --no-cpu-baseline --problem final-13682 --step 2 --steps 5 --warmup 1 --repeats 3"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['e0_layout']; print(round(d['value']), 'terms/s', round(d['graph_us_per_term'],1), 'us per term;', c.get('camera_chunks_step2'), 'kernel', c.get('e0_kernel_step2'), c.get('e0_tune_us_step2'))"; }
run() { label=$1; shift; echo -n "$label: " | tee -a $out/summary.txt; env "$@" $B 2> $out/err.txt | line | tee -a $out/summary.txt; }
run "library's choice, library's stride" A=1
run "library's choice, 1536            " POVAR_CKH_STRIDE=1536
run "e0_ck_h forced, library's stride  " POVAR_E0_CK=1
run "e0_ck_h forced, 1536              " POVAR_E0_CK=1 POVAR_CKH_STRIDE=1536
