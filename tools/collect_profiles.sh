#!/bin/bash
# Copies what tools/round_evidence.sh <tag> left under gpurun_out/<tag> into profiles/ (tracked) and stamps
# profiles/traffic.json with the hash of the kernel sources:  tools/collect_profiles.sh <tag>
set -eu
tag=${1:-r02}
cd "$(dirname "$0")/.."
R=gpurun_out/$tag; P=profiles
cp $R/kt/*/*kernel_stats.csv $P/${tag}_kernel_stats_default_cmd.csv
cp $R/sq/*/*counter_collection.csv $P/${tag}_pmc_sq.csv
cp $R/fetch/*/*counter_collection.csv $P/${tag}_pmc_fetch_size.csv
cp $R/write/*/*counter_collection.csv $P/${tag}_pmc_write_size.csv
cp $R/summary.json $P/${tag}_profile_summary.json
cp $R/bench_default.json $P/${tag}_bench.json
cp $R/kt_bench.json $P/${tag}_bench_under_trace.json
for n in step2 huber ladybug trafalgar final_huber; do cp $R/bench_$n.json $P/${tag}_bench_$n.json; done
cp $R/popularity.txt $P/${tag}_popularity_sweep.txt
cp $R/shards.txt $P/${tag}_shard_term_times.txt
cp $R/bal_venice/kernel_stats.txt $P/${tag}_bal_venice_kernel_stats.txt
(echo "# tools/run_bal_config.py <problem> --power-sc-iterations 20 (both LM steps, defaults otherwise); venice: 6 + 4 iterations, --eta 0"
 cat $R/bal_ladybug-49.json $R/bal_trafalgar-257.json $R/bal_venice/bal_summary.json) > $P/${tag}_bal_end_to_end.txt
python3 - > $P/${tag}_bal_venice_stage_times_ms.txt <<'PY'
import json
d = json.load(open("gpurun_out/ba_log_venice-1778.json"))
for k in ("iteration_time", "jacobian_evaluation_time", "prepare_time", "solve_reduced_system_time",
          "back_substitution_time", "residual_evaluation_time"):
    print(k, [round(x * 1e3, 3) for x in d[k]])
PY
python3 tools/pmc_to_traffic.py $R/fetch/*/*counter_collection.csv $R/write/*/*counter_collection.csv venice-1778:ldsacc:1 $P/traffic.json
