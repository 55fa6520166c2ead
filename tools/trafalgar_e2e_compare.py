#!/usr/bin/env python3
"""bin/bal (HIP) vs build/bal_oracle (CPU twin) on the trafalgar-257 shape, step 1 run to function tolerance, then
step 2: prints the accept/reject sequences and the relative cost differences per iteration.
usage: trafalgar_e2e_compare.py [extra bal flags...]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from povar_amd import synth  # noqa: E402

name = os.environ.get("PROBLEM", "trafalgar-257")
n_c, n_l, n_o = synth.BAL_SHAPES[name]
path = os.path.join(tempfile.gettempdir(), f"problem-{n_c}-{n_l}-pre.txt")
if not os.path.exists(path):
    synth.write_data_custom(path, synth.make_bal_problem(name))
extra = sys.argv[1:]
logs = {}
for binary, tag in (("bin/bal", "hip"), ("build/bal_oracle", "oracle")):
    log = os.path.join(tempfile.gettempdir(), f"{tag}.json")
    r = subprocess.run([os.path.join(ROOT, binary), "--input", path, "--log-log-path", log, "--quiet"] + extra, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    logs[tag] = json.load(open(log))
a, b = logs["hip"], logs["oracle"]
print("iterations", len(a["cost"]), len(b["cost"]))
n = min(len(a["cost"]), len(b["cost"]))
ca, cb = np.array(a["cost"][:n]), np.array(b["cost"][:n])
for i in range(n):
    print(a["iteration"][i], "%.10e %.10e rel %.2e" % (ca[i], cb[i], abs(ca[i] / cb[i] - 1)), a["step_is_successful"][i], b["step_is_successful"][i],
          a["linear_solver_iterations"][i], b["linear_solver_iterations"][i])
