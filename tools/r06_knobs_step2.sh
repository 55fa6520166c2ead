#!/bin/bash
# knob sweep of e0_ck_h on the two-batch layout (bench.py --step 2 --steps 100 x 3, one process each)
out=gpurun_out/${1:-r06_knobs2}; mkdir -p $out; rm -f $out/summary.txt
run() { echo -n "$* : " | tee -a $out/summary.txt; env "$@" python3 bench.py --no-cpu-baseline --no-secondary --step 2 --steps 100 --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['e0_layout']['camera_chunks_step2']; print(round(d['value']), round(d['graph_us_per_term'],2), c['batches'], c['chunks'], c['own_record_chunks'], c['accumulators'])" | tee -a $out/summary.txt; }
run POVAR_E0_CK=1
run POVAR_E0_CK=1 POVAR_CK_TILE_COST=6
run POVAR_E0_CK=1 POVAR_CK_TILE_COST=20
run POVAR_E0_CK=1 POVAR_CK_TILE_COST=32
run POVAR_E0_CK=1 POVAR_CK_HMAX=10
run POVAR_E0_CK=1 POVAR_CK_HMAX=12
run POVAR_E0_CK=1 POVAR_CK_NOPLACE=1
run POVAR_E0_CK=1 POVAR_CKH_ACC_CAP=280
run POVAR_E0_CK=1 POVAR_CKH_ACC_CAP=240
run POVAR_E0_CK=1
