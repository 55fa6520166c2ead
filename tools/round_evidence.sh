#!/bin/bash
# Everything the round's profiles/ directory is built from, in one GPU session:  tools/round_evidence.sh <tag>
#   gpurun_out/<tag>/{kt,sq,fetch,write,summary.json}   tools/profile_e0.sh (default bench command under rocprofv3)
#   gpurun_out/<tag>/bench_*.json                        plain bench lines (headline with the CPU baseline, step 2, HUBER,
#                                                        the other BASELINE shapes)
#   gpurun_out/<tag>/popularity.txt, shards.txt          sensitivity to the synthetic graph; per-rank term time of a shard
#   gpurun_out/<tag>/bal_*                               bin/bal end to end + its per-kernel trace (venice)
set -u
tag=${1:-r02}
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out   # (locally: also delete gpurun_out/<tag> before merging a new run, run ids differ)
tools/profile_e0.sh $tag > $out/profile.log 2>&1
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --step 2 --no-cpu-baseline --no-secondary > $out/bench_step2.json 2> /dev/null
python3 bench.py --robust-norm HUBER --no-cpu-baseline --no-secondary > $out/bench_huber.json 2> /dev/null
python3 bench.py --problem ladybug-49 --no-cpu-baseline --no-secondary > $out/bench_ladybug.json 2> /dev/null
python3 bench.py --problem trafalgar-257 --no-cpu-baseline --no-secondary > $out/bench_trafalgar.json 2> /dev/null
python3 bench.py --problem final-13682 --robust-norm HUBER --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > $out/bench_final_huber.json 2> /dev/null
tools/popularity_sweep.sh $out/popularity > $out/popularity.txt 2>&1
tools/shard_sweep.sh > $out/shards.txt 2>&1
for p in ladybug-49 trafalgar-257; do
  python3 tools/run_bal_config.py $p --power-sc-iterations 20 > $out/bal_$p.json 2> $out/bal_$p.err
done
tools/bal_kernel_trace.sh $tag/bal_venice venice-1778 --max-num-iterations-step-1 6 --max-num-iterations-step-2 4 --power-sc-iterations 20 --eta 0 > $out/bal_venice.log 2>&1
rm -rf $out/bal_venice/kt/*/*kernel_trace.csv  # large; the stats table is kept
ls -la $out
