#!/bin/bash
# Profiles of the bench command for one round tag:  tools/profile_e0.sh <tag> [extra bench args]
#   gpurun_out/<tag>/kt     rocprofv3 --kernel-trace --stats          -> per-kernel average durations
#   gpurun_out/<tag>/sq     --pmc SQ_* (own pass)                     -> wave-cycle breakdown
#   gpurun_out/<tag>/fetch, gpurun_out/<tag>/write  --pmc FETCH_SIZE / WRITE_SIZE (separate passes, MI355X_MICROARCH.md)
# rocprofv3 is given the program itself after `--` (python3 bench.py ...), no wrapper.
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-secondary $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B --steps 10 --warmup 2 > $out/kt_bench.json 2> $out/kt.err
export POVAR_NO_GRAPH=1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq -- $B --steps 2 --warmup 1 > /dev/null 2> $out/sq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 3 --warmup 1 > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 3 --warmup 1 > /dev/null 2> $out/write.err
python3 tools/profile_summary.py $out
