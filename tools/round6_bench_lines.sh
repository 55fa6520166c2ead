#!/bin/bash
# The bench lines of the round-6 table, AFTER profiles/traffic.json carries the PMC bytes of the committed kernel sources
# (tools/collect_profiles_r06.sh r06 r06T traffic), so that every line's roofline is on measured bytes; + the kernel trace of the
# default command.  tools/round6_bench_lines.sh [tag] -> gpurun_out/<tag>/bench_*.json, bench_kt/
set -u
tag=${1:-r06B}
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py"
$B > $out/bench_default.json 2> $out/bench_default.err < /dev/null
$B --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> /dev/null < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_kt -- python3 bench.py --no-cpu-baseline --no-secondary --warm-seconds 0 --repeats 1 > $out/bench_kt.json 2> $out/bench_kt.err < /dev/null
rm -rf $out/bench_kt/*/*kernel_trace.csv
B="python3 bench.py --no-secondary --no-cpu-baseline"
POVAR_E0_CK=0 $B > $out/bench_forced_e0_lpl.json 2> /dev/null < /dev/null
$B --robust-norm HUBER > $out/bench_huber.json 2> /dev/null < /dev/null
$B --popularity local > $out/bench_local.json 2> /dev/null < /dev/null
$B --popularity zipf0.5 > $out/bench_zipf05.json 2> /dev/null < /dev/null
$B --popularity uniform > $out/bench_uniform.json 2> /dev/null < /dev/null
$B --problem trafalgar-257 > $out/bench_trafalgar.json 2> /dev/null < /dev/null
POVAR_RES=0 $B --problem trafalgar-257 > $out/bench_trafalgar_per_term_kernels.json 2> /dev/null < /dev/null
$B --problem ladybug-49 > $out/bench_ladybug.json 2> /dev/null < /dev/null
POVAR_RES=0 $B --problem ladybug-49 > $out/bench_ladybug_per_term_kernels.json 2> /dev/null < /dev/null
$B --step 2 > $out/bench_step2.json 2> /dev/null < /dev/null
POVAR_E0_CK=0 $B --step 2 > $out/bench_step2_forced_e0_lpl_h.json 2> /dev/null < /dev/null
$B --step 2 --robust-norm HUBER > $out/bench_step2_huber.json 2> /dev/null < /dev/null
$B --problem final-13682 --robust-norm HUBER --huber 20 --steps 5 --warmup 1 > $out/bench_final_huber.json 2> /dev/null < /dev/null
POVAR_DETERMINISTIC=1 $B --steps 40 > $out/bench_deterministic.json 2> /dev/null < /dev/null
POVAR_DETERMINISTIC=1 POVAR_DET_CK=0 $B --steps 40 > $out/bench_deterministic_gather.json 2> /dev/null < /dev/null
POVAR_DETERMINISTIC=1 $B --steps 40 --robust-norm HUBER > $out/bench_deterministic_huber.json 2> /dev/null < /dev/null
POVAR_DETERMINISTIC=1 $B --steps 40 --step 2 > $out/bench_deterministic_step2.json 2> /dev/null < /dev/null
POVAR_DETERMINISTIC=1 POVAR_DET_CK=0 $B --steps 40 --step 2 > $out/bench_deterministic_step2_gather.json 2> /dev/null < /dev/null
ls $out
