#!/bin/bash
# After a change that touches ONE kernel (tools/isa_same.py says which): the PMC passes of the cases that run it again, traffic.json
# stamped on the box with the hash of the sources that are running, then every bench line of the table.
#   tools/r06_final_lines.sh "<case> <case> ..."     cases: ck1 huber_ck1 local_ck1 zipf05_ck1 uniform_ck1 step2_ckh    -> gpurun_out/r06T/pmc_<case>, gpurun_out/r06B/
set -u
cd "$(dirname "$0")/.." || exit 1
newest() { ls -t $@ | head -1; }
for n in ${1:-}; do
  case $n in
    huber_ck1) bash tools/round6_traffic_one.sh r06T huber_ck1 1 --robust-norm HUBER ;;
    ck1) bash tools/round6_traffic_one.sh r06T ck1 1 ;;
    local_ck1) bash tools/round6_traffic_one.sh r06T local_ck1 1 --popularity local ;;
    zipf05_ck1) bash tools/round6_traffic_one.sh r06T zipf05_ck1 1 --popularity zipf0.5 ;;
    uniform_ck1) bash tools/round6_traffic_one.sh r06T uniform_ck1 1 --popularity uniform ;;
    step2_ckh) bash tools/round6_traffic_one.sh r06T step2_ckh 1 --step 2 ;;
    *) echo "unknown case $n"; exit 1 ;;
  esac
  cp $(newest gpurun_out/r06T/pmc_$n/fetch/*/*counter_collection.csv) profiles/r06_pmc_fetch_size_$n.csv
  cp $(newest gpurun_out/r06T/pmc_$n/write/*/*counter_collection.csv) profiles/r06_pmc_write_size_$n.csv
done
bash tools/stamp_traffic_r06.sh > gpurun_out/r06B_stamp.log 2>&1
cp profiles/traffic.json gpurun_out/traffic_stamped_on_box.json
bash tools/round6_bench_lines.sh r06B
