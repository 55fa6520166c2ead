#!/bin/bash
# The plain bench lines of the round (no profiler):  tools/bench_lines.sh <out dir>
# Run AFTER profiles/traffic.json carries the PMC bytes of the current kernel sources (tools/collect_profiles_r03.sh), so
# that every line's roofline is on measured bytes.
set -u
out=$1
cd "$(dirname "$0")/.." || exit 1
mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --step 2 --no-cpu-baseline --no-secondary > $out/bench_step2.json 2> /dev/null
python3 bench.py --robust-norm HUBER --no-cpu-baseline --no-secondary > $out/bench_huber.json 2> /dev/null
python3 bench.py --problem ladybug-49 --no-cpu-baseline --no-secondary > $out/bench_ladybug.json 2> /dev/null
python3 bench.py --problem trafalgar-257 --no-cpu-baseline --no-secondary > $out/bench_trafalgar.json 2> /dev/null
python3 bench.py --problem final-13682 --robust-norm HUBER --huber 20 --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > $out/bench_final_huber.json 2> /dev/null
python3 bench.py --popularity local --no-cpu-baseline --no-secondary > $out/bench_local.json 2> /dev/null
python3 bench.py --problem final-13682 --popularity local --robust-norm HUBER --huber 20 --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > $out/bench_final_local_huber.json 2> /dev/null
