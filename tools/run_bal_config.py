#!/usr/bin/env python3
"""Runs the restated `bal` program end to end on a synthetic problem with a BAL shape
(BASELINE.json configs 1-3): writes the data_custom file, runs bin/bal, summarises ba_log.json.

usage: run_bal_config.py <problem> [--synth-init-gt] [extra bal flags...]      e.g. trafalgar-257 --power-sc-iterations 20
(--synth-init-gt: initial cameras = the generator's ground truth perturbed by 2 %, a start inside the basin of both steps,
instead of the reference's random initial cameras)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from povar_amd import synth  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    gt = "--synth-init-gt" in extra
    extra = [e for e in extra if e != "--synth-init-gt"]
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    n_c, n_l, n_o = synth.BAL_SHAPES[name]
    import tempfile
    path = os.path.join(tempfile.gettempdir(), f"problem-{n_c}-{n_l}-{'gt' if gt else 'pre'}.txt")  # 200 MB for venice: not into gpurun_out
    if not os.path.exists(path):
        synth.write_data_custom(path, synth.make_bal_problem(name, init="gt", init_noise=0.02) if gt else synth.make_bal_problem(name))
    log = os.path.join(out_dir, f"ba_log_{name}{'_gpus' + extra[extra.index('--gpus') + 1] if '--gpus' in extra else ''}.json")
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "bin", "bal"), "--input", path, "--log-log-path", log, "--quiet"] + extra,
                       capture_output=True, text=True)
    wall = time.time() - t0
    open(os.path.join(out_dir, f"bal_{name}.stdout"), "w").write(r.stdout)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-2000:])
        sys.exit(r.returncode)
    d = json.load(open(log))
    n1 = [i for i, it in enumerate(d["iteration"]) if it == 0][1]
    solve = d["solve_reduced_system_time"]
    iters = d["linear_solver_iterations"]
    terms = sum(iters)
    summary = {
        "problem": name, "start": "ground truth perturbed by 2 %" if gt else "random initial cameras (the reference's)",
        "flags": extra, "wall_s": round(wall, 2),
        "costs_step1": d["cost"][:n1], "costs_step2": d["cost"][n1:],
        "accepted_flags": [int(x) for x in d["step_is_successful"]],
        "solver": d["_static"]["solver"],
        "step1": {"iterations": n1 - 1, "cost_first": d["cost"][0], "cost_last": d["cost"][n1 - 1],
                  "accepted": int(sum(d["step_is_successful"][1:n1]))},
        "step2": {"iterations": len(d["cost"]) - n1 - 1, "cost_first": d["cost"][n1], "cost_last": d["cost"][-1],
                  "accepted": int(sum(d["step_is_successful"][n1 + 1:]))},
        "power_series_terms": terms, "solve_reduced_system_time_s": round(sum(solve), 4),
        "time_per_term_us": round(1e6 * sum(solve) / max(terms, 1), 1),
        "prepare_time_s": round(sum(d["prepare_time"]), 4), "stage1_time_s": round(sum(d["stage1_time"]), 4),
        "back_substitution_time_s": round(sum(d["back_substitution_time"]), 4),
    }
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
