#!/bin/bash
# Round 6, VERDICT r05 item 5: would e0_ck_h with TWO landmark batches (2048 slots of 64 bytes, ~ 310 accumulators) beat three
# batches (1536 slots, 501 accumulators) on venice?  A build with CKH_STRIDE = 2048 (build/libpovar_hip_ckh2048.so: a sed of two
# constants in a copy of the sources) against the shipped library, accumulators capped through POVAR_HOT_ACC.
out=gpurun_out/${1:-r06_ckh}; mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --no-secondary --no-cpu-baseline --repeats 3 --step 2"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']['e0_layout']; print(round(d['value']), 'terms/s', round(d['graph_us_per_term'],2), 'us per term;', c.get('camera_chunks_step2'), 'acc', c['lds_camera_slots'], 'kernel', c.get('e0_kernel_step2'), c.get('e0_tune_us_step2'))"; }
run() { label=$1; shift; echo -n "$label: " | tee -a $out/summary.txt; env "$@" $B 2> $out/err.txt | line | tee -a $out/summary.txt; }
run "shipped (1536 slots), 501 accumulators" POVAR_E0_CK=1
run "shipped (1536 slots), 400 accumulators" POVAR_E0_CK=1 POVAR_HOT_ACC=400
run "shipped (1536 slots), 310 accumulators" POVAR_E0_CK=1 POVAR_HOT_ACC=310
run "2048 slots, 310 accumulators          " POVAR_E0_CK=1 POVAR_HOT_ACC=310 POVAR_LIB=$PWD/build/libpovar_hip_ckh2048.so
run "2048 slots, 280 accumulators          " POVAR_E0_CK=1 POVAR_HOT_ACC=280 POVAR_LIB=$PWD/build/libpovar_hip_ckh2048.so
run "2048 slots, 501 accumulators          " POVAR_E0_CK=1 POVAR_LIB=$PWD/build/libpovar_hip_ckh2048.so
