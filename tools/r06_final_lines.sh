#!/bin/bash
# After the last host-side change (HUBER keeps cold records): the HUBER PMC passes again, traffic.json stamped on the box with the
# hash of the sources that are running, then every bench line of the table.  -> gpurun_out/r06T/pmc_huber_ck1, gpurun_out/r06B/
set -u
cd "$(dirname "$0")/.." || exit 1
bash tools/round6_traffic_one.sh r06T huber_ck1 1 --robust-norm HUBER
newest() { ls -t $@ | head -1; }
cp $(newest gpurun_out/r06T/pmc_huber_ck1/fetch/*/*counter_collection.csv) profiles/r06_pmc_fetch_size_huber_ck1.csv
cp $(newest gpurun_out/r06T/pmc_huber_ck1/write/*/*counter_collection.csv) profiles/r06_pmc_write_size_huber_ck1.csv
bash tools/stamp_traffic_r06.sh > gpurun_out/r06B_stamp.log 2>&1
cp profiles/traffic.json gpurun_out/traffic_stamped_on_box.json
bash tools/round6_bench_lines.sh r06B
