#!/bin/bash
# knob sweep of e0_ck on the final round-6 kernel (bench.py --steps 100 x 3, one process each)
out=gpurun_out/${1:-r06_knobs}; mkdir -p $out; rm -f $out/summary.txt
run() { echo -n "$* : " | tee -a $out/summary.txt; env "$@" python3 bench.py --no-cpu-baseline --no-secondary --steps 100 --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['graph_us_per_term'],2), round(1e3*d['kernel_ms']['e0'],2), d['config']['e0_layout']['camera_chunks']['batches'])" | tee -a $out/summary.txt; }
run POVAR_E0_CK=1
run POVAR_E0_CK=2
run POVAR_E0_CK=1 POVAR_CK_TILE_COST=8
run POVAR_E0_CK=1 POVAR_CK_TILE_COST=20
run POVAR_E0_CK=1 POVAR_CK_HMAX=12
run POVAR_E0_CK=1 POVAR_CK_NOPLACE=1
run POVAR_E0_CK=1 POVAR_LPL_PLACE=none
run POVAR_E0_CK=1
