#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs, POVAR_NO_GRAPH=1) of the timed term loops of bench.py, one case per key
# of profiles/traffic.json, the step-1 E0 kernel FORCED in every case (under the counters' serialisation the library's
# own timing of the two kernels is not the one of a normal run):  tools/round5_traffic.sh [tag]  -> gpurun_out/<tag>/pmc_<case>/
set -u
tag=${1:-r06T}
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
pmc_case() {  # name, E0 kernel (POVAR_E0_CK), bench args...
  local name=$1 ck=$2; shift 2
  local B="python3 bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 $*"
  mkdir -p $out/pmc_$name
  POVAR_E0_CK=$ck POVAR_NO_GRAPH=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_$name/fetch -- $B > /dev/null 2> $out/pmc_$name/fetch.err < /dev/null
  POVAR_E0_CK=$ck POVAR_NO_GRAPH=1 timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_$name/write -- $B > /dev/null 2> $out/pmc_$name/write.err < /dev/null
}
pmc_case ck1 1
pmc_case lpl 0
pmc_case huber 0 --robust-norm HUBER
pmc_case huber_ck1 1 --robust-norm HUBER
pmc_case local_ck1 1 --popularity local
pmc_case local 0 --popularity local
pmc_case zipf05_ck1 1 --popularity zipf0.5
pmc_case uniform_ck1 1 --popularity uniform
pmc_case step2 0 --step 2
pmc_case step2_ckh 1 --step 2
POVAR_DETERMINISTIC=1 pmc_case det 1
POVAR_DETERMINISTIC=1 pmc_case det_step2 1 --step 2
pmc_case final_huber 0 --problem final-13682 --robust-norm HUBER --huber 20
pmc_case final_local_huber 0 --problem final-13682 --popularity local --robust-norm HUBER --huber 20
ls $out
