mkdir -p gpurun_out/r03d
for w in 256 512 1024; do
  POVAR_E0_WGS=$w timeout 900 python bench.py --problem final-13682 --robust-norm HUBER --huber 20 --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > gpurun_out/r03d/final_wgs$w.json 2> gpurun_out/r03d/final_wgs$w.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03d/final_wgs$w.json"))
print($w, round(d["value"],1), "terms/s", d["kernel_ms"], d["config"]["e0_layout"])
PY
done
