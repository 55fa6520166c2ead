#!/bin/bash
# profiles/traffic.json from the PMC passes kept under profiles/ (r06_pmc_fetch_size_*.csv, r06_pmc_write_size_*.csv), stamped
# with the hash of the current kernel sources; tools/collect_profiles_r06.sh calls it after copying new passes in.
set -eu
cd "$(dirname "$0")/.."
P=profiles
t() { python3 tools/pmc_to_traffic.py $P/r06_pmc_fetch_size_$1.csv $P/r06_pmc_write_size_$1.csv $2 $P/traffic.json; }
t ck1 venice-1778:ldsacc:1:ck1
t det venice-1778:ldsacc:1:ck7
t det_step2 venice-1778:ldsacc:1:step2:ckh2
t lpl venice-1778:ldsacc:1
t huber venice-1778:ldsacc:1:HUBER
t huber_ck1 venice-1778:ldsacc:1:HUBER:ck1
t local_ck1 venice-1778:ldsacc:1:local:ck1
t local venice-1778:ldsacc:1:local
t zipf05_ck1 venice-1778:ldsacc:1:zipf0.5:ck1
t uniform_ck1 venice-1778:ldsacc:1:uniform:ck1
t step2 venice-1778:ldsacc:1:step2
t step2_ckh venice-1778:ldsacc:1:step2:ckh1
t final_huber final-13682:ldsacc:1:HUBER
t final_local_huber final-13682:ldsacc:1:HUBER:local
