#!/bin/bash
# Measured-byte rooflines of every kernel of an LM iteration: three rocprofv3 passes over tools/stage_loop.py (kernel
# trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- separate passes, POVAR_NO_GRAPH=1 so the term kernels are plain launches)
#   tools/stage_rooflines.sh <out dir under gpurun_out> <stage_loop.py args...>   -> <out>/{kt,fetch,write}, <out>/stages.json
set -u
out=gpurun_out/$1; shift
cd "$(dirname "$0")/.." || exit 1
mkdir -p $out
export TMPDIR=/tmp POVAR_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/stage_loop.py "$@" --iters 6 > $out/kt.out 2> $out/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/stage_loop.py "$@" --iters 2 > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/stage_loop.py "$@" --iters 2 > /dev/null 2> $out/write.err
python3 tools/stage_table.py $out "$@"
rm -rf $out/kt/*/*kernel_trace.csv $out/kt/*/*agent_info.csv
