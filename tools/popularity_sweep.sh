#!/bin/bash
# Sensitivity of the headline to the synthetic graph: terms/s and LDS-resident fraction on the
# venice-1778 shape for other camera popularity laws and with a long-track tail.  usage: tools/popularity_sweep.sh <outdir>
out=${1:-gpurun_out/popularity}
mkdir -p $out
cd "$(dirname "$0")/.." || exit 1
for v in "zipf1 0" "zipf0.5 0" "uniform 0" "local 0" "zipf1 0.1" "zipf1 0.25" "zipf0.5 0.1"; do
  set -- $v
  python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --popularity $1 --long-track-frac $2 > $out/pop_$1_$2.json 2> $out/pop_$1_$2.err
  python3 - <<PY
import json
d = json.load(open("$out/pop_$1_$2.json"))
l = d["config"]["e0_layout"]
print("popularity=$1 long_track_frac=$2: %.0f terms/s, E0 %.1f us, LDS-resident %.1f %%, global %d + grid %d cameras, rows %d"
      % (d["value"], d["kernel_ms"]["e0"] * 1e3, 100 * l["lds_resident_obs_frac"], l["global_cameras"], l["grid_cameras"], l["rows"]))
PY
done
POVAR_E0_V1=1 python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --popularity uniform | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-1 kernel (POVAR_E0_V1=1), uniform: %.0f terms/s' % d['value'])"
POVAR_E0_V1=1 python3 bench.py --no-cpu-baseline --no-secondary --steps 10 --popularity zipf0.5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-1 kernel (POVAR_E0_V1=1), zipf0.5: %.0f terms/s' % d['value'])"
POVAR_E0_V1=1 python3 bench.py --no-cpu-baseline --no-secondary --steps 10 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-1 kernel (POVAR_E0_V1=1), zipf1: %.0f terms/s' % d['value'])"
