#!/bin/bash
# Round-6 evidence in one GPU session:  tools/round5_evidence.sh [tag]     -> gpurun_out/<tag>/...
#   e0/          tools/ck_pmc.sh: kernel trace, FETCH_SIZE, WRITE_SIZE, SQ passes of the venice term loop with e0_lpl (0) and e0_ck (1)
#   e0_huber/    the same with the HUBER norm
#   step2_kt/    kernel trace of the step-2 term loop with e0_lpl_h and e0_ck_h (tools/ckh_trace.py)
#   stamps.txt   in-kernel phase stamps of e0_ck (diagnostic build: tools/variants/build_variant.sh ck_stamps stamps)
#   res_*        the resident power series: per-term times against the per-term kernels (tools/res_term_time.py), in-kernel
#                phase stamps (tools/variants/res_stamps.py + tools/res_stamps_report.py), kernel trace of both forms
#   det_*        POVAR_DETERMINISTIC=1: tools/det_probe.py (time per term, bit identity, distance from the default mode), kernel trace
#   bench_*.json plain bench lines (the library's own kernel choice unless the name says otherwise)
#   sweep.txt    e0_ck against e0_lpl over the graph families (tools/ck_sweep.sh)
#   shards.txt   sharded term times on one GPU (tools/shard_sweep.sh)
#   bal_*        bin/bal end to end
set -u
tag=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
bash tools/ck_pmc.sh $out/e0 venice-1778 0,1 > $out/e0_summary.txt 2>&1 < /dev/null
bash tools/ck_pmc.sh $out/e0_huber "venice-1778 --robust HUBER" 0,1 > $out/e0_huber_summary.txt 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step2_kt -- python3 tools/ckh_trace.py venice-1778 --variants 0,1 > $out/step2_kt.out 2>&1 < /dev/null
rm -rf $out/step2_kt/*/*kernel_trace.csv
POVAR_LIB=build/libpovar_hip_stamps.so POVAR_E0_CK=1 timeout 300 python3 tools/ck_stamps.py venice-1778 --variant 1 > $out/stamps.txt 2>&1 < /dev/null
# the resident power series
(for a in "ladybug-49 1" "trafalgar-257 1" "trafalgar-257 1 HUBER" "venice-1778 16" "venice-1778 32"; do timeout 200 python3 tools/res_term_time.py $a 2>&1 | tail -1; done
 for a in "venice-1778 8" "venice-1778 4"; do POVAR_RES_MAX_OBS=2000000 timeout 200 python3 tools/res_term_time.py $a 2>&1 | tail -1; done) > $out/res_term_times.txt 2>&1 < /dev/null
(for a in "ladybug-49 1" "trafalgar-257 1" "venice-1778 16"; do timeout 200 python3 tools/res_stamps_report.py $a 2>&1 | sed -E "s/at median +[0-9]+ +min +[0-9]+ +max +[0-9]+//"; done
 POVAR_RES_MAX_OBS=2000000 timeout 200 python3 tools/res_stamps_report.py venice-1778 8 2>&1 | sed -E "s/at median +[0-9]+ +min +[0-9]+ +max +[0-9]+//") > $out/res_stamps.txt 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/res_kt -- python3 tools/res_term_time.py trafalgar-257 > $out/res_kt.out 2>&1 < /dev/null
rm -rf $out/res_kt/*/*kernel_trace.csv
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_kt -- python3 bench.py --no-cpu-baseline --no-secondary --warm-seconds 0 --repeats 1 > $out/bench_kt.json 2> $out/bench_kt.err < /dev/null
rm -rf $out/bench_kt/*/*kernel_trace.csv
B="python3 bench.py"
$B > $out/bench_default.json 2> $out/bench_default.err < /dev/null
$B --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> /dev/null < /dev/null
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
(for a in "venice-1778 NONE" "venice-1778 HUBER" "trafalgar-257 NONE" "final-13682 HUBER 5"; do timeout 600 python3 tools/det_probe.py $a 2>&1; done) > $out/det_probe.txt < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/det_kt -- python3 tools/det_probe.py venice-1778 NONE 5 > $out/det_kt.out 2>&1 < /dev/null
rm -rf $out/det_kt/*/*kernel_trace.csv
bash tools/ck_sweep.sh $out/sweep_raw.txt 1 > $out/sweep.txt 2>&1 < /dev/null
bash tools/shard_sweep.sh > $out/shards.txt 2>&1 < /dev/null
(echo "# rank 0's shard of final-13682 HUBER at world = 8 (what one rank of BASELINE config 5 executes)"; POVAR_FORCE_COMM=1 python3 tools/shard_term_time.py 8 final-13682 2>&1 | grep "world=") >> $out/shards.txt 2>&1 < /dev/null
python3 tools/run_bal_config.py venice-1778 --max-num-iterations-step-1 6 --max-num-iterations-step-2 4 --power-sc-iterations 20 --eta 0 > $out/bal_venice.json 2> $out/bal_venice.err < /dev/null
python3 tools/run_bal_config.py venice-1778 --max-num-iterations-step-1 30 --max-num-iterations-step-2 0 --power-sc-iterations 20 --eta 0 > $out/bal_venice_step1.json 2> $out/bal_venice_step1.err < /dev/null
python3 tools/run_bal_config.py venice-1778 --synth-init-gt --max-num-iterations-step-1 8 --max-num-iterations-step-2 6 --power-sc-iterations 20 > $out/bal_venice_gt.json 2> $out/bal_venice_gt.err < /dev/null
python3 tools/run_bal_config.py trafalgar-257 --synth-init-gt --max-num-iterations-step-1 30 --max-num-iterations-step-2 30 --power-sc-iterations 20 --eta 0 > $out/bal_trafalgar.json 2> $out/bal_trafalgar.err < /dev/null
python3 tools/run_bal_config.py ladybug-49 --synth-init-gt --max-num-iterations-step-1 30 --max-num-iterations-step-2 30 --power-sc-iterations 20 --eta 0 > $out/bal_ladybug.json 2> $out/bal_ladybug.err < /dev/null
rm -rf $out/e0/kt/*/*kernel_trace.csv $out/e0_huber/kt/*/*kernel_trace.csv
ls $out
