#!/bin/bash
# Round-3 evidence in one GPU session:  tools/round3_evidence.sh [tag]     -> gpurun_out/<tag>/...
#   e0/         tools/profile_e0.sh passes (kernel trace, SQ, FETCH_SIZE, WRITE_SIZE) on the default bench command
#   e0_<case>/  the same PMC passes for the other timed term loops: step 2, HUBER, final-13682 HUBER, the local graphs
#   stages_*/   tools/stage_rooflines.sh: every kernel of an LM iteration (venice step 1 / step 2, final-13682 HUBER step 1)
#   bench_*.json plain bench lines; popularity.txt; shards.txt; create.txt; bal_* end to end
set -u
tag=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
tools/profile_e0.sh $tag/e0 > $out/profile_e0.log 2>&1
pmc_case() {  # name, bench args...
  local name=$1; shift
  local B="python3 bench.py --no-cpu-baseline --no-secondary $*"
  mkdir -p $out/e0_$name
  POVAR_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/e0_$name/fetch -- $B --steps 2 --warmup 1 > /dev/null 2> $out/e0_$name/fetch.err
  POVAR_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/e0_$name/write -- $B --steps 2 --warmup 1 > /dev/null 2> $out/e0_$name/write.err
}
pmc_case step2 --step 2
pmc_case huber --robust-norm HUBER
pmc_case final_huber --problem final-13682 --robust-norm HUBER --huber 20
pmc_case local --popularity local
pmc_case final_local_huber --problem final-13682 --popularity local --robust-norm HUBER --huber 20
tools/stage_rooflines.sh $tag/stages_venice_step1 venice-1778 --step 1 > $out/stages_venice_step1.md 2>&1
tools/stage_rooflines.sh $tag/stages_venice_step2 venice-1778 --step 2 > $out/stages_venice_step2.md 2>&1
tools/stage_rooflines.sh $tag/stages_final_huber_step1 final-13682 --step 1 --robust-norm HUBER --huber 20 > $out/stages_final_huber_step1.md 2>&1
tools/bench_lines.sh $out   # (lines on model bytes if traffic.json is stale: re-run tools/bench_lines.sh after collecting)
tools/popularity_sweep.sh $out/popularity > $out/popularity.txt 2>&1
tools/shard_sweep.sh > $out/shards.txt 2>&1
python3 tools/create_time.py venice-1778 final-13682 > $out/create.txt 2>&1
for p in ladybug-49 trafalgar-257; do
  python3 tools/run_bal_config.py $p --power-sc-iterations 20 > $out/bal_$p.json 2> $out/bal_$p.err
done
# venice end to end: as a caller gets it (rows placed on a host thread meanwhile: a ten-iteration run is over before they
# arrive), then with the placement inside povar_create (POVAR_LPL_PLACE=sync: the steady state of a long run) under the trace
python3 tools/run_bal_config.py venice-1778 --max-num-iterations-step-1 6 --max-num-iterations-step-2 4 --power-sc-iterations 20 --eta 0 > $out/bal_venice_default.json 2> $out/bal_venice_default.err
cp gpurun_out/ba_log_venice-1778.json $out/ba_log_venice_default.json
POVAR_LPL_PLACE=sync tools/bal_kernel_trace.sh $tag/bal_venice venice-1778 --max-num-iterations-step-1 6 --max-num-iterations-step-2 4 --power-sc-iterations 20 --eta 0 > $out/bal_venice.log 2>&1
cp gpurun_out/ba_log_venice-1778.json $out/ba_log_venice_sync.json
rm -rf $out/bal_venice/kt/*/*kernel_trace.csv $out/e0/kt/*/*kernel_trace.csv
ls $out
