#!/usr/bin/env python3
"""Where does the pipeline reach the generator's noise floor?  Runs the `bal` program (default: the oracle-backed twin
build/bal_oracle; --binary bin/bal for the HIP library) over solver routes / alpha / power-series orders on seeded
synthetic problems and prints one table row per run: final step-1 pOSE cost, first and final step-2 cost, final
mean residual [px], the chi-square floor 1/2 sigma^2 (2 n_obs - 11 n_c - 3 n_l + 15) and its 5-sigma verdict.

usage: known_answer_sweep.py [--binary build/bal_oracle] [--set small|p49|ladybug|gt]   (output -> profiles/r03_known_answer_sweep.txt)"""
import argparse
import math
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from povar_amd import synth  # noqa: E402


def run(binary, f, flags):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, binary), "--input", f, "--log-log-path", os.path.join(tempfile.gettempdir(), "ka.json")] + flags,
                       capture_output=True, text=True)
    fin = re.findall(r"Final Cost: error: (\S+) \(mean res: (\S+),", r.stdout)
    it0 = re.findall(r"Iteration 0, error: (\S+) \(mean res: (\S+),", r.stdout)
    n_it = len(re.findall(r"\[(Success|Reject|Invalid)\]", r.stdout))
    return fin, it0, n_it, time.time() - t0


ROUTES = [
    # step 1, step 2, m, extra flags
    ("CHOLESKY", "RIPCG", 20, []),
    ("CHOLESKY", "RIPOBA", 20, []),
    ("POWER_VARPROJ", "RIPCG", 20, []),
    ("POWER_VARPROJ", "RIPOBA", 20, []),
    ("POWER_VARPROJ", "RIPOBA", 20, ["--eta", "0"]),
    ("POWER_VARPROJ", "RIPOBA", 50, ["--eta", "0"]),
    ("POWER_VARPROJ", "RIPOBA", 500, ["--eta", "0"]),
    ("POWER_SCHUR_COMPLEMENT", "RIPOBA", 20, []),
    ("PCG", "RIPOBA", 20, []),
    ("CHOLESKY", "RIPCG", 20, ["--alpha", "0.1"]),
    ("CHOLESKY", "RIPOBA", 20, ["--alpha", "0.1"]),
    ("POWER_VARPROJ", "RIPOBA", 10, ["--alpha", "0.1"]),
    ("POWER_VARPROJ", "RIPOBA", 20, ["--alpha", "0.1"]),
    ("POWER_SCHUR_COMPLEMENT", "RIPOBA", 20, ["--alpha", "0.1"]),
    ("PCG", "RIPCG", 20, ["--alpha", "0.1"]),
]
SETS = {
    "small": [((10, 300, 1300), 21, {}), ((10, 300, 1300), 3, {}), ((20, 600, 3000), 5, {})],
    "p49": [((49, 2000, 8200), 7, {})],
    "ladybug": [("ladybug-49", None, {})],
    "gt": [((49, 2000, 8200), 7, {"init": "gt", "init_noise": 0.05}), ("trafalgar-257", None, {"init": "gt", "init_noise": 0.02})],
    "venice": [("venice-1778", None, {"init": "gt", "init_noise": 0.02})],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--binary", default="build/bal_oracle")
    ap.add_argument("--set", default="small", choices=list(SETS))
    ap.add_argument("--step2-iterations", default="300")
    a = ap.parse_args()
    print(f"# binary {a.binary}; --max-num-iterations-step-1 100 --max-num-iterations-step-2 {a.step2_iterations}; noise 0.5 px")
    for shape, seed, kw in SETS[a.set]:
        p = synth.make_bal_problem(shape, **kw) if isinstance(shape, str) else synth.make_problem(*shape, seed=seed, **kw)
        f = os.path.join(tempfile.gettempdir(), f"ka-{p.n_cams}-{p.n_lms}-{seed}.txt")
        synth.write_data_custom(f, p)
        dof = 2 * p.n_obs - (11 * p.n_cams + 3 * p.n_lms - 15)
        exp, std = 0.125 * dof, 0.125 * math.sqrt(2 * dof)
        print(f"\n## {p.n_cams} cams / {p.n_lms} landmarks / {p.n_obs} obs, seed {seed}, {kw or 'random initial cameras'}: "
              f"chi-square floor {exp:.1f} +- {std:.1f}")
        print(f"{'step 1':<23}{'step 2':<8}{'m':>4}  {'flags':<16}{'pOSE final':>12}{'step-2 start':>14}{'step-2 final':>14}{'mean px':>9}{'LM its':>7}  floor?")
        routes = ROUTES if not kw else [r for r in ROUTES if r[0].startswith("POWER") and r[1] == "RIPOBA" and r[2] == 20 and "--eta" not in r[3]]
        for s1, s2, m, extra in routes:
            fin, it0, n_it, dt = run(a.binary, f, ["--solver-type-step-1", s1, "--solver-type-step-2", s2, "--power-sc-iterations", str(m),
                                                   "--max-num-iterations-step-1", "100", "--max-num-iterations-step-2", a.step2_iterations] + extra)
            if len(fin) < 2 or len(it0) < 2:
                print(f"{s1:<23}{s2:<8}{m:>4}  {' '.join(extra):<16}  run failed")
                continue
            c2 = float(fin[1][0])
            print(f"{s1:<23}{s2:<8}{m:>4}  {' '.join(extra):<16}{float(fin[0][0]):>12.4e}{float(it0[1][0]):>14.4e}{c2:>14.4e}{float(fin[1][1]):>9.2f}{n_it:>7}  "
                  f"{'yes' if abs(c2 - exp) <= 5 * std else 'NO'}", flush=True)


if __name__ == "__main__":
    main()
