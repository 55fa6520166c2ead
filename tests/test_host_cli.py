"""CPU checks of the restated host surface (loader, CLI, LM loops) via the oracle-backed twin of
`bal` (tests/cpp/bal_oracle.cpp): no GPU involved."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binaries():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "povar_amd", "csrc"), "host"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "bin", "bal"), os.path.join(ROOT, "build", "bal_oracle")


def test_cli_flags_and_errors(binaries):
    bal, _ = binaries
    r = subprocess.run([bal, "--help"], capture_output=True, text=True)
    for flag in ("--solver-type-step-1", "--solver-type-step-2", "--power-sc-iterations", "--max-num-iterations-step-1",
                 "--residual-robust-norm", "--residual-huber-parameter", "--eta", "--r-tolerance", "--alpha",
                 "--initial-trust-region-radius", "--log-log-path", "--create-dataset", "--num-threads", "--deterministic"):
        assert flag in r.stdout, flag
    assert subprocess.run([bal], capture_output=True).returncode != 0                      # missing --input
    assert subprocess.run([bal, "--input", "x", "--bogus", "1"], capture_output=True).returncode != 0
    assert subprocess.run([bal, "--input", "x", "--solver-type-step-1", "NOPE"], capture_output=True).returncode != 0
    assert subprocess.run([bal, "--input", "/nonexistent/file"], capture_output=True).returncode != 0


def test_lm_loop_cost_decreases_and_log(binaries, tmp_path):
    from povar_amd import synth
    _, bal_oracle = binaries
    p = synth.make_problem(8, 150, 640, seed=5)
    f = str(tmp_path / "problem-8-150.txt")
    synth.write_data_custom(f, p)
    log = str(tmp_path / "ba_log.json")
    r = subprocess.run([bal_oracle, "--input", f, "--log-log-path", log, "--quiet", "--max-num-iterations-step-1", "30",
                        "--max-num-iterations-step-2", "0", "--power-sc-iterations", "20"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(log))
    n1 = [i for i, it in enumerate(d["iteration"]) if it == 0][1]
    cost = np.array(d["cost"][:n1])
    ok = np.array(d["step_is_successful"][:n1], dtype=bool)
    acc = cost[ok]
    assert np.all(np.diff(acc) < 0) and acc[-1] < 0.05 * acc[0]      # accepted steps decrease the pOSE cost
    assert "Iteration 0, error:" in r.stdout and "[Success]" in r.stdout and "Final Cost:" in r.stdout
    assert d["_type"] == "rootba_povar" and d["_static"]["problem_info"]["num_observations"] == p.n_obs


def test_create_dataset_roundtrip(binaries, tmp_path):
    """--create-dataset: original BAL text -> data_custom/<name> with seeded random cameras
    (bal_problem.cpp:307-471), readable by load_bal_eccv."""
    from povar_amd import synth
    bal, bal_oracle = binaries
    p = synth.make_problem(5, 40, 140, seed=8)
    lm_of = np.repeat(np.arange(p.n_lms), np.diff(p.lm_off))
    src = tmp_path / "problem-5-40-pre.txt"
    with open(src, "w") as fh:
        fh.write(f"{p.n_cams} {p.n_lms} {p.n_obs}\n")
        for c, l, (u, v) in zip(p.cam_idx, lm_of, p.obs):
            fh.write(f"{c} {l} {u:.6f} {-v:.6f}\n")
        for c in range(p.n_cams):
            fh.write("\n".join(["0.1", "0.2", "0.3", "0", "0", "-5", "800", "0", "0"]) + "\n")
        for x in p.lms.ravel():
            fh.write(f"{x:.6f}\n")
    r = subprocess.run([bal, "--input", str(src), "--create-dataset", "--random-seed", "7"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "data_custom" / "problem-5-40-pre.txt"
    q = synth.read_data_custom(str(out))
    assert (q.n_cams, q.n_lms, q.n_obs) == (p.n_cams, p.n_lms, p.n_obs)
    assert np.abs(q.obs - p.obs).max() < 1e-9 and np.array_equal(q.cam_idx, p.cam_idx)
    assert np.allclose(q.cams[:, 8:], [0, 0, 0, 1]) and np.abs(q.cams[:, :8]).max() > 0.1
    # same seed -> same file
    first = open(out).read()
    subprocess.run([bal, "--input", str(src), "--create-dataset", "--random-seed", "7"], cwd=tmp_path, capture_output=True)
    assert open(out).read() == first
    r2 = subprocess.run([bal_oracle, "--input", str(out), "--quiet", "--max-num-iterations-step-1", "2",
                         "--max-num-iterations-step-2", "0", "--log-log-path", str(tmp_path / "l.json")], capture_output=True, text=True)
    assert r2.returncode == 0


@pytest.mark.parametrize("step1", ["PCG", "CHOLESKY"])
def test_lm_loop_explicit_sc_solvers(binaries, tmp_path, step1):
    """--solver-type-step-1 PCG | CHOLESKY (LinearizorSC) through the same LM loop: the exact solve and the
    truncated PCG both drive the pOSE cost down, and the log names the explicit-SC linear solver."""
    from povar_amd import synth
    bal, bal_oracle = binaries
    r = subprocess.run([bal, "--help"], capture_output=True, text=True)
    assert "--max-linear-solver-iterations" in r.stdout and "--min-linear-solver-iterations" in r.stdout
    p = synth.make_problem(8, 150, 640, seed=5)
    f = str(tmp_path / "problem-8-150.txt")
    synth.write_data_custom(f, p)
    log = str(tmp_path / "ba_log.json")
    r = subprocess.run([bal_oracle, "--input", f, "--log-log-path", log, "--quiet", "--solver-type-step-1", step1,
                        "--solver-type-step-2", "RIPCG", "--max-num-iterations-step-1", "20",
                        "--max-num-iterations-step-2", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(log))
    n1 = [i for i, it in enumerate(d["iteration"]) if it == 0][1]
    cost = np.array(d["cost"][:n1])
    ok = np.array(d["step_is_successful"][:n1], dtype=bool)
    acc = cost[ok]
    assert np.all(np.diff(acc) < 0) and acc[-1] < 0.05 * acc[0]
    its = d["linear_solver_iterations"][1:n1]
    assert all(i == 0 for i in its) if step1 == "CHOLESKY" else all(0 < i <= 500 for i in its)
    assert d["_static"]["solver"]["solver_type"] == ("bal_pcg" if step1 == "PCG" else "variable_projection")
