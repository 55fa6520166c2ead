#!/bin/bash
# Copies what tools/round5_evidence.sh <tag> and tools/round5_traffic.sh <ttag> left under gpurun_out/ into profiles/
# (tracked) and stamps profiles/traffic.json (one key per measured term loop) with the hash of the kernel sources:
#   tools/collect_profiles_r06.sh [tag] [ttag] [traffic]
set -eu
tag=${1:-r06}; ttag=${2:-r06T}
cd "$(dirname "$0")/.."
R=gpurun_out/$tag; T=gpurun_out/$ttag; P=profiles
newest() { ls -t $@ | head -1; }
if [ "${3:-all}" != "traffic" ]; then  # (third argument "traffic": only the PMC passes -> profiles/traffic.json)
cp $(newest $R/e0/kt/*/*kernel_stats.csv) $P/r06_kernel_stats_e0_lpl_vs_e0_ck.csv
cp $(newest $R/e0/sq/*/*counter_collection.csv) $P/r06_pmc_sq_e0_lpl_vs_e0_ck.csv
cp $R/e0_summary.txt $P/r06_e0_lpl_vs_e0_ck_summary.txt
cp $(newest $R/e0_huber/kt/*/*kernel_stats.csv) $P/r06_kernel_stats_e0_lpl_vs_e0_ck_huber.csv
cp $R/e0_huber_summary.txt $P/r06_e0_lpl_vs_e0_ck_huber_summary.txt
cp $(newest $R/step2_kt/*/*kernel_stats.csv) $P/r06_kernel_stats_step2_e0_lpl_h_vs_e0_ck_h.csv
cp $(newest $R/res_kt/*/*kernel_stats.csv) $P/r06_kernel_stats_series_res_trafalgar.csv
grep -v '^  XCD ' $R/stamps.txt > $P/r06_e0_ck_phase_stamps.txt   # (per-XCD start spreads: the CUs' s_memtime counters are not synchronised)
cp $R/res_term_times.txt $P/r06_res_term_times.txt
cp $R/res_stamps.txt $P/r06_res_phase_stamps.txt
cp $R/sweep.txt $P/r06_e0_ck_graph_families.txt
cp $R/shards.txt $P/r06_shard_term_times.txt
cp $R/det_probe.txt $P/r06_det_probe.txt
cp $(newest $R/det_kt/*/*kernel_stats.csv) $P/r06_kernel_stats_deterministic.csv
cp $R/bench_default.json $P/r06_bench.json
if ls $R/bench_kt/*/*kernel_stats.csv > /dev/null 2>&1; then cp $(newest $R/bench_kt/*/*kernel_stats.csv) $P/r06_kernel_stats_bench_default_cmd.csv; fi   # rocprofv3 --kernel-trace --stats of the bench command
for n in driver_flags forced_e0_lpl huber local zipf05 uniform trafalgar trafalgar_per_term_kernels ladybug ladybug_per_term_kernels step2 step2_forced_e0_lpl_h step2_huber final_huber deterministic deterministic_gather deterministic_huber deterministic_step2 deterministic_step2_gather; do cp $R/bench_$n.json $P/r06_bench_$n.json; done
(echo "# tools/run_bal_config.py venice-1778 --max-num-iterations-step-1 6 --max-num-iterations-step-2 4 --power-sc-iterations 20 --eta 0"
 cat $R/bal_venice.json
 echo "# venice-1778, step 1 only, 30 iterations x 20 terms (--eta 0)"
 cat $R/bal_venice_step1.json
 echo "# venice-1778 from the ground truth perturbed by 2 % (--synth-init-gt)"
 cat $R/bal_venice_gt.json
 echo "# trafalgar-257 (BASELINE config 3) and ladybug-49 (config 2) from the perturbed ground truth, --eta 0: step 1's series runs as the resident kernel"
 cat $R/bal_trafalgar.json $R/bal_ladybug.json) > $P/r06_bal_end_to_end.txt
fi
if [ -d $T ]; then
for n in ck1 det det_step2 lpl huber huber_ck1 local_ck1 local zipf05_ck1 uniform_ck1 step2 step2_ckh final_huber final_local_huber; do
  cp $(newest $T/pmc_$n/fetch/*/*counter_collection.csv) $P/r06_pmc_fetch_size_$n.csv
  cp $(newest $T/pmc_$n/write/*/*counter_collection.csv) $P/r06_pmc_write_size_$n.csv
done
bash tools/stamp_traffic_r06.sh
fi
