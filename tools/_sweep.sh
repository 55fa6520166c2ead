run() { echo -n "$* : "; env "$@" python3 bench.py --no-cpu-baseline --no-secondary --steps 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['kernel_ms']['e0'],2), d['config']['e0_layout']['e0_tune_us'], d['config']['e0_layout']['camera_chunks']['batches'])"; }
run POVAR_E0_CK=1
run POVAR_E0_CK=2
run POVAR_E0_CK=4
run POVAR_E0_CK=5
run POVAR_E0_CK=1 POVAR_CK_NB=3
run POVAR_E0_CK=1 POVAR_CK_NB=4
run POVAR_E0_CK=4 POVAR_CK_NB=4
run POVAR_E0_CK=1 POVAR_CK_TILE_COST=20
run POVAR_E0_CK=1 POVAR_CK_HMAX=12
