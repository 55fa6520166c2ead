#!/bin/bash
# one case of tools/round6_traffic.sh again:  tools/round6_traffic_one.sh <tag> <name> <E0 kernel> [bench args...]
set -u
tag=$1; name=$2; ck=$3; shift 3
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag; mkdir -p $out/pmc_$name
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --warm-seconds 0 --repeats 1 $*"
POVAR_E0_CK=$ck POVAR_NO_GRAPH=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_$name/fetch -- $B > /dev/null 2> $out/pmc_$name/fetch.err < /dev/null
POVAR_E0_CK=$ck POVAR_NO_GRAPH=1 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_$name/write -- $B > /dev/null 2> $out/pmc_$name/write.err < /dev/null
