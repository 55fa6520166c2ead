#!/bin/bash
# Round-4 evidence in one GPU session:  tools/round4_evidence.sh [tag]     -> gpurun_out/<tag>/...
#   e0/          tools/ck_pmc.sh: kernel trace, FETCH_SIZE, WRITE_SIZE, SQ passes of the venice term loop with e0_lpl (0) and e0_ck (1)
#   e0_huber/    the same with the HUBER norm
#   step2_kt/    kernel trace of the step-2 term loop with e0_lpl_h and e0_ck_h (tools/ckh_trace.py)
#   stamps.txt   in-kernel phase stamps of e0_ck (diagnostic build, tools/ck_stamps.py)
#   bench_*.json plain bench lines (the library's own kernel choice unless the name says otherwise)
#   sweep.txt    e0_ck against e0_lpl over the graph families (tools/ck_sweep.sh)
#   shards.txt   sharded term times on one GPU (tools/shard_sweep.sh)
#   bal_*        bin/bal end to end: one context, and --gpus 2 (two shard contexts of one process on this device)
set -u
tag=${1:-r04}
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
export TMPDIR=/tmp
bash tools/ck_pmc.sh $out/e0 venice-1778 0,1 > $out/e0_summary.txt 2>&1 < /dev/null
bash tools/ck_pmc.sh $out/e0_huber "venice-1778 --robust HUBER" 0,1 > $out/e0_huber_summary.txt 2>&1 < /dev/null
# step 2: e0_lpl_h and e0_ck_h in one process under the kernel trace
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/step2_kt -- python3 tools/ckh_trace.py venice-1778 --variants 0,1 > $out/step2_kt.out 2>&1 < /dev/null
rm -rf $out/step2_kt/*/*kernel_trace.csv
POVAR_LIB=build/libpovar_hip_stamps.so POVAR_E0_CK=1 timeout 300 python3 tools/ck_stamps.py venice-1778 --variant 1 > $out/stamps.txt 2>&1 < /dev/null
B="python3 bench.py --no-secondary"
$B > $out/bench_default.json 2> $out/bench_default.err < /dev/null
POVAR_E0_CK=0 $B --no-cpu-baseline > $out/bench_forced_e0_lpl.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --robust-norm HUBER > $out/bench_huber.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --popularity local > $out/bench_local.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --popularity zipf0.5 > $out/bench_zipf05.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --popularity uniform > $out/bench_uniform.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --problem trafalgar-257 > $out/bench_trafalgar.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --problem ladybug-49 > $out/bench_ladybug.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --step 2 > $out/bench_step2.json 2> /dev/null < /dev/null
POVAR_E0_CK=0 $B --no-cpu-baseline --step 2 > $out/bench_step2_forced_e0_lpl_h.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --step 2 --robust-norm HUBER > $out/bench_step2_huber.json 2> /dev/null < /dev/null
$B --no-cpu-baseline --problem final-13682 --robust-norm HUBER --huber 20 --steps 5 --warmup 1 > $out/bench_final_huber.json 2> /dev/null < /dev/null
bash tools/ck_sweep.sh $out/sweep_raw.txt 1 > $out/sweep.txt 2>&1 < /dev/null
bash tools/shard_sweep.sh > $out/shards.txt 2>&1 < /dev/null
python3 tools/run_bal_config.py venice-1778 --max-num-iterations-step-1 6 --max-num-iterations-step-2 4 --power-sc-iterations 20 --eta 0 > $out/bal_venice.json 2> $out/bal_venice.err < /dev/null
python3 tools/run_bal_config.py venice-1778 --max-num-iterations-step-1 6 --max-num-iterations-step-2 4 --power-sc-iterations 20 --eta 0 --gpus 2 > $out/bal_venice_gpus2.json 2> $out/bal_venice_gpus2.err < /dev/null
for g in 1 2; do
  python3 tools/run_bal_config.py venice-1778 --synth-init-gt --max-num-iterations-step-1 8 --max-num-iterations-step-2 6 --power-sc-iterations 20 --gpus $g > $out/bal_venice_gt_gpus$g.json 2> $out/bal_venice_gt_gpus$g.err < /dev/null
done
rm -rf $out/e0/kt/*/*kernel_trace.csv $out/e0_huber/kt/*/*kernel_trace.csv
ls $out
