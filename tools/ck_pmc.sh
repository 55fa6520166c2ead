#!/bin/bash
# Kernel durations and HBM bytes of the step-1 E0 kernels on one problem:  tools/ck_pmc.sh <out dir> <problem> <variants> [env assignments ...]
#   <out>/kt     rocprofv3 --kernel-trace --stats     (term loop of every variant in turn: tools/ck_trace.py)
#   <out>/fetch, <out>/write   --pmc FETCH_SIZE / WRITE_SIZE, separate passes, POVAR_NO_GRAPH=1
#   <out>/sq     --pmc SQ_* (own pass)
out=$1; prob=$2; vars=$3; shift 3
for a in "$@"; do export "$a"; done
mkdir -p $out
export TMPDIR=/tmp
T="python3 tools/ck_trace.py $prob --variants $vars"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $T > $out/kt.out 2> $out/kt.err < /dev/null
export POVAR_NO_GRAPH=1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $T --solves 1 > /dev/null 2> $out/fetch.err < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $T --solves 1 > /dev/null 2> $out/write.err < /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq -- $T --solves 1 > /dev/null 2> $out/sq.err < /dev/null
python3 tools/ck_pmc_summary.py $out < /dev/null
