#!/bin/bash
# Copies what tools/round3_evidence.sh <tag> left under gpurun_out/<tag> into profiles/ (tracked) and stamps
# profiles/traffic.json (one key per measured term loop) with the hash of the kernel sources:  tools/collect_profiles_r03.sh [tag]
set -eu
tag=${1:-r03}
cd "$(dirname "$0")/.."
R=gpurun_out/$tag; P=profiles
# gpurun merges a new run into gpurun_out/ without deleting what an earlier run of the same tag left there (the
# rocprofv3 output files carry the process id): always take the newest match
newest() { ls -t $@ | head -1; }
cp $(newest $R/e0/kt/*/*kernel_stats.csv) $P/${tag}_kernel_stats_default_cmd.csv
cp $(newest $R/e0/sq/*/*counter_collection.csv) $P/${tag}_pmc_sq.csv
cp $(newest $R/e0/fetch/*/*counter_collection.csv) $P/${tag}_pmc_fetch_size.csv
cp $(newest $R/e0/write/*/*counter_collection.csv) $P/${tag}_pmc_write_size.csv
cp $R/e0/summary.json $P/${tag}_profile_summary.json
cp $R/e0/kt_bench.json $P/${tag}_bench_under_trace.json
cp $R/bench_default.json $P/${tag}_bench.json
for n in step2 huber ladybug trafalgar final_huber local final_local_huber; do cp $R/bench_$n.json $P/${tag}_bench_$n.json; done
for n in step2 huber final_huber local final_local_huber; do
  cp $(newest $R/e0_$n/fetch/*/*counter_collection.csv) $P/${tag}_pmc_fetch_size_$n.csv
  cp $(newest $R/e0_$n/write/*/*counter_collection.csv) $P/${tag}_pmc_write_size_$n.csv
done
for n in venice_step1 venice_step2 final_huber_step1; do
  cp $R/stages_$n.md $P/${tag}_stages_$n.md
  cp $R/stages_$n/stages.json $P/${tag}_stages_$n.json
done
cp $R/popularity.txt $P/${tag}_popularity_sweep.txt
cp $R/shards.txt $P/${tag}_shard_term_times.txt
grep -E "povar_create|build_lpl|row placement" $R/create.txt > $P/${tag}_create_time.txt
cp $R/bal_venice/kernel_stats.txt $P/${tag}_bal_venice_kernel_stats.txt
(echo "# tools/run_bal_config.py <problem> --power-sc-iterations 20 (both LM steps, defaults otherwise); venice: 6 + 4 iterations, --eta 0"
 cat $R/bal_ladybug-49.json $R/bal_trafalgar-257.json
 echo "# venice as a caller gets it: povar_create returns on the natural row order, the placed rows would arrive after ~0.4 s"
 cat $R/bal_venice_default.json
 echo "# venice with POVAR_LPL_PLACE=sync (placement inside povar_create: the steady state of a long run)"
 cat $R/bal_venice/bal_summary.json) > $P/${tag}_bal_end_to_end.txt
python3 - $R > $P/${tag}_bal_venice_stage_times_ms.txt <<'PY'
import json, sys
for title, f in (("POVAR_LPL_PLACE=sync (placed rows: the steady state)", "ba_log_venice_sync.json"),
                 ("default (natural row order: the run ends before the placed rows arrive)", "ba_log_venice_default.json")):
    d = json.load(open(sys.argv[1] + "/" + f))
    print("#", title)
    for k in ("iteration_time", "jacobian_evaluation_time", "prepare_time", "solve_reduced_system_time",
              "back_substitution_time", "residual_evaluation_time"):
        print(k, [round(x * 1e3, 3) for x in d[k]])
PY
t() { python3 tools/pmc_to_traffic.py $(newest $R/$1/fetch/*/*counter_collection.csv) $(newest $R/$1/write/*/*counter_collection.csv) $2 $P/traffic.json; }
t e0 venice-1778:ldsacc:1
t e0_step2 venice-1778:ldsacc:1:step2
t e0_huber venice-1778:ldsacc:1:HUBER
t e0_final_huber final-13682:ldsacc:1:HUBER
t e0_local venice-1778:ldsacc:1:local
t e0_final_local_huber final-13682:ldsacc:1:HUBER:local
