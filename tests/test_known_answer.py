"""Known-answer runs of the whole pipeline (loader -> LM/VarPro step 1 -> projective step 2) on synthetic
problems whose ground truth the generator knows (povar_amd/synth.py: pinhole cameras, 0.5 px Gaussian noise).

This is the only check in the repo that does not go through the builder's own restatements: whatever the
residuals, Jacobians, the 3x3 eliminations, the power series, the back-substitution and the LM loops
(bal_bundle_adjustment.cpp:252-843) compute, a run that converges must end at the maximum-likelihood cost of the
noise,  E[cost] = 1/2 sigma^2 (2 n_obs - p),  p = 11 n_cams + 3 n_lms - 15  (projective gauge),
with standard deviation 1/2 sigma^2 sqrt(2 (2 n_obs - p))  (chi-square).  A wrong sign, scale, index or
Jacobian anywhere on the path leaves the cost orders of magnitude above that.

CPU half: the oracle-backed twin of `bal` (tests/cpp/bal_oracle.cpp).  tests/test_gpu_known_answer.py runs the
same configurations through bin/bal (HIP library).

What reaches the floor and what does not (tools/known_answer_sweep.py, profiles/r03_known_answer_sweep.txt,
DESIGN.md section 8): from a start inside the basin every route does, at every size; from the reference's random
initial cameras it depends on the route and the graph (exact solves included) because step 1 can only raise
lambda (the l_diff quirk, test below) -- the configurations parametrised here are ones that do."""
import json
import math
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOISE_PX = 0.5


def chi2_floor(n_cams, n_lms, n_obs, sigma=NOISE_PX):
    dof = 2 * n_obs - (11 * n_cams + 3 * n_lms - 15)
    return 0.5 * sigma ** 2 * dof, 0.5 * sigma ** 2 * math.sqrt(2 * dof)


def run_bal(binary, path, log, flags, timeout=900):
    cmd = [os.path.join(ROOT, binary), "--input", path, "--log-log-path", log] + list(flags)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.load(open(log))
    fin = [(float(a), float(b)) for a, b in re.findall(r"Final Cost: error: (\S+) \(mean res: (\S+),", r.stdout)]
    start = [float(a) for a in re.findall(r"Iteration 0, error: (\S+) ", r.stdout)]
    n1 = [i for i, it in enumerate(d["iteration"]) if it == 0]
    return {"log": d, "final": fin, "start": start, "n1": n1[1] if len(n1) > 1 else len(d["iteration"]), "stdout": r.stdout}


def write_problem(tmp_path, shape, seed):
    from povar_amd import synth
    p = synth.make_problem(*shape, seed=seed) if not isinstance(shape, str) else synth.make_bal_problem(shape)
    f = str(tmp_path / f"problem-{p.n_cams}-{p.n_lms}.txt")
    synth.write_data_custom(f, p)
    return p, f


# (problem, seed, flags): every one of these must end on the noise floor.
KNOWN_ANSWER = [
    # exact reduced solves both steps: pins residuals/Jacobians/elimination/back-substitution/LM against geometry
    pytest.param((10, 300, 1300), 21, ["--solver-type-step-1", "CHOLESKY", "--solver-type-step-2", "RIPCG"], id="p10-cholesky-ripcg"),
    pytest.param((49, 2000, 8200), 7, ["--solver-type-step-1", "CHOLESKY", "--solver-type-step-2", "RIPCG", "--alpha", "0.1"],
                 id="p49-cholesky-ripcg"),
    # step-2 power series (RIPOBA, m = 20) from the exact step-1 result
    pytest.param((10, 300, 1300), 21, ["--solver-type-step-1", "CHOLESKY", "--solver-type-step-2", "RIPOBA", "--power-sc-iterations", "20"],
                 id="p10-cholesky-ripoba20"),
    # the hot path end to end with the README's settings (README.md:75-83)
    pytest.param((10, 300, 1300), 21, ["--solver-type-step-1", "POWER_VARPROJ", "--solver-type-step-2", "RIPOBA", "--power-sc-iterations", "20",
                                       "--alpha", "0.1"], id="p10-varproj20-ripoba20"),
    pytest.param((10, 300, 1300), 3, ["--solver-type-step-1", "POWER_VARPROJ", "--solver-type-step-2", "RIPOBA", "--power-sc-iterations", "20",
                                      "--alpha", "0.1"], id="p10b-varproj20-ripoba20"),
    pytest.param((20, 600, 3000), 5, ["--solver-type-step-1", "POWER_VARPROJ", "--solver-type-step-2", "RIPOBA", "--power-sc-iterations", "20",
                                      "--alpha", "0.1"], id="p20-varproj20-ripoba20"),
    pytest.param((10, 300, 1300), 21, ["--solver-type-step-1", "POWER_SCHUR_COMPLEMENT", "--solver-type-step-2", "RIPOBA",
                                       "--power-sc-iterations", "20"], id="p10-powersc20-ripoba20"),
    # the README's example command (POWER_SCHUR_COMPLEMENT + RIPOBA) on the 49-camera BASELINE shape
    pytest.param("ladybug-49", None, ["--solver-type-step-1", "POWER_SCHUR_COMPLEMENT", "--solver-type-step-2", "RIPOBA",
                                      "--power-sc-iterations", "20", "--alpha", "0.1"], id="ladybug49-powersc20-ripoba20"),
]
COMMON = ["--max-num-iterations-step-1", "100", "--max-num-iterations-step-2", "300"]


def check_floor(p, res):
    exp, std = chi2_floor(p.n_cams, p.n_lms, p.n_obs)
    cost, mean_res = res["final"][-1]
    # the generator's floor: mean |r| of 2-d Gaussian noise, shrunk by the fitted degrees of freedom
    floor_px = NOISE_PX * math.sqrt(math.pi / 2) * math.sqrt(2 * exp / NOISE_PX ** 2 / (2 * p.n_obs))
    assert abs(cost - exp) <= 5 * std, (cost, exp, std)
    assert abs(mean_res / floor_px - 1) <= 0.10, (mean_res, floor_px)
    assert res["log"]["_static"]["solver"]["termination_type"] == "CONVERGENCE"


@pytest.fixture(scope="module")
def bal_oracle():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "povar_amd", "csrc"), "host"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    return "build/bal_oracle"


@pytest.mark.parametrize("shape,seed,flags", KNOWN_ANSWER)
def test_pipeline_reaches_noise_floor(bal_oracle, tmp_path, shape, seed, flags):
    p, f = write_problem(tmp_path, shape, seed)
    res = run_bal(bal_oracle, f, str(tmp_path / "log.json"), flags + COMMON + ["--quiet"])
    check_floor(p, res)


def test_varproj_model_decrease_quirk_doubles_lambda(bal_oracle, tmp_path):
    """landmark_block.hpp:699-705 multiplies the UNSCALED fresh Jp with the SCALED increment (linearizor_power_varproj.cpp:
    250-256 re-scales `inc` before back_substitute_pOSE): |l_diff| is inflated by the squared Jacobian column norms,
    so rho = f_diff / l_diff is a tiny negative number on every step and `lambda *= 1 - (2 rho - 1)^3`
    (bal_bundle_adjustment.cpp:461) doubles lambda on EVERY accepted VarPro step -- step 1 of the reference ends by
    function tolerance after ~25 accepted steps whatever the linear solver (the same holds for CHOLESKY / PCG:
    linearizor_sc.cpp:70-88).  Restated as is; this test documents the consequence."""
    p, f = write_problem(tmp_path, (10, 300, 1300), 21)
    res = run_bal(bal_oracle, f, str(tmp_path / "log.json"),
                  ["--solver-type-step-1", "POWER_VARPROJ", "--power-sc-iterations", "20", "--max-num-iterations-step-1", "100",
                   "--max-num-iterations-step-2", "0", "--quiet"])
    d, n1 = res["log"], res["n1"]
    ok = np.array(d["step_is_successful"][1:n1], dtype=bool)
    rho = np.array(d["relative_decrease"][1:n1])
    tr = np.array(d["trust_region_radius"][:n1])
    assert ok.all() and np.all(rho < 0) and np.all(rho > -1e-5)
    assert np.allclose(tr[1:] / tr[:-1], 0.5, rtol=1e-4)          # lambda doubles, accepted or not
    assert 20 <= n1 <= 40 and "Function tolerance reached" in res["stdout"]


def test_random_start_stall_is_the_route_not_the_algebra(bal_oracle, tmp_path):
    """From the reference's RANDOM initial cameras the outcome depends on the route (profiles/r03_known_answer_sweep.txt:
    exact CHOLESKY + RIPCG runs miss the floor on some graphs too, and more terms are not monotonically better): step 1
    ends after ~25 accepted steps wherever it is (lambda can only grow, test above) and step 2 inherits that basin.  One
    instance, pinned here: with the code default alpha = 0.01 (solver_options.hpp:129; README.md:83 says 0.1) the m = 20
    series leaves step 1 where step 2 stalls at ~19 px; the SAME code with m = 500 terms ends on the noise floor, as do
    the exact solve and POWER_SCHUR_COMPLEMENT (parametrised test above).  So the stall is the truncated series (spectral
    radius ~1 - lambda, SURVEY 8c: 0.99997 at lambda = 1e-4) under that lambda schedule -- not an error in the algebra
    the HIP path shares with the oracle.  From a start inside the basin (synth init="gt") every power route reaches the
    floor at every size tried (tests/test_gpu_baseline_sizes.py: trafalgar-257, venice-1778)."""
    p, f = write_problem(tmp_path, (10, 300, 1300), 21)
    common = ["--solver-type-step-2", "RIPOBA", "--max-num-iterations-step-1", "100", "--max-num-iterations-step-2", "300", "--quiet"]
    short = run_bal(bal_oracle, f, str(tmp_path / "a.json"), ["--solver-type-step-1", "POWER_VARPROJ", "--power-sc-iterations", "20"] + common)
    long = run_bal(bal_oracle, f, str(tmp_path / "b.json"),
                   ["--solver-type-step-1", "POWER_VARPROJ", "--power-sc-iterations", "500", "--eta", "0"] + common)
    assert short["final"][-1][1] > 5.0            # px: far from the floor (measured 18.9)
    check_floor(p, long)


@pytest.mark.parametrize("shape,seed,noise", [((49, 2000, 8200), 7, 0.05), ((20, 600, 3000), 5, 0.05)])
@pytest.mark.parametrize("step1", ["POWER_VARPROJ", "POWER_SCHUR_COMPLEMENT"])
def test_power_routes_reach_floor_from_inside_the_basin(bal_oracle, tmp_path, shape, seed, noise, step1):
    """Ground-truth cameras perturbed by 5 % (synth init="gt"), code-default alpha: both power-series routes of step 1
    followed by RIPOBA (m = 20 each) end on the chi-square floor -- on the 49-camera graph where NO route gets there
    from the random start."""
    from povar_amd import synth
    p = synth.make_problem(*shape, seed=seed, init="gt", init_noise=noise)
    f = str(tmp_path / "gt.txt")
    synth.write_data_custom(f, p)
    res = run_bal(bal_oracle, f, str(tmp_path / "log.json"),
                  ["--solver-type-step-1", step1, "--solver-type-step-2", "RIPOBA", "--power-sc-iterations", "20", "--quiet"] + COMMON)
    check_floor(p, res)
