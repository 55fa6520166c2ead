"""ba_log.json written by the drop-in surface carries the reference's full schema (bal/ba_log.hpp:54-245,
ba_log.cpp:63-150, ba_log_utils.cpp:43-186) and loads with a reader that does what python/rootba/log.py does
(json.load -> lists of numbers become arrays, mappings become attribute dictionaries).  The reader below is a
fixture-style mirror of that file's key accesses, not a copy of it.  Runs on CPU: the oracle-backed twin of `bal`
(tests/cpp/bal_oracle.cpp) writes the log through the same host code as bin/bal."""
import json
import os
import subprocess
from numbers import Number

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# BaLog::BaIteration, ba_log.hpp:147-245
PER_ITERATION = {
    "iteration": int, "linear_solver_type": str, "step_is_valid": bool, "step_is_nonmonotonic": bool,
    "step_is_successful": bool, "num_obs": int, "num_obs_valid": int, "num_obs_valid_change": int, "cost": float,
    "cost_change": float, "cost_valid": float, "cost_valid_change": float, "cost_avg_valid": float,
    "cost_avg_valid_change": float, "grad_projected_norm": float, "grad_projected_max_norm": float, "grad_norm": float,
    "grad_max_norm": float, "residual_block_mean": float, "residual_block_valid_mean": float, "step_norm": float,
    "relative_decrease": float, "trust_region_radius": float, "linear_solver_iterations": int, "iteration_time": float,
    "cumulative_time": float, "logging_time": float, "step_solver_time": float, "residual_evaluation_time": float,
    "jacobian_evaluation_time": float, "scale_landmark_jacobian_time": float, "perform_qr_time": float,
    "stage1_time": float, "scale_pose_jacobian_time": float, "landmark_damping_time": float,
    "compute_preconditioner_time": float, "compute_gradient_time": float, "stage2_time": float, "prepare_time": float,
    "solve_reduced_system_time": float, "back_substitution_time": float, "update_cameras_time": float,
    "resident_memory": int, "resident_memory_peak": int,
}
# BaLog::ProblemInfo / PipelineTiming / BaSolver, ba_log.hpp:56-144
PROBLEM_INFO = ["type", "input_path", "num_cameras", "num_landmarks", "num_observations", "rcs_sparsity", "per_lm_obs",
                "per_host_lms"]
TIMING = ["total", "load", "preprocess", "optimize", "postprocess"]
SOLVER = ["solver_type", "termination_type", "message", "num_successful_steps", "num_unsuccessful_steps",
          "logging_time_in_seconds", "grouping_time_in_seconds", "preprocessor_time_in_seconds",
          "minimizer_time_in_seconds", "postprocessor_time_in_seconds", "total_time_in_seconds",
          "linear_solver_time_in_seconds", "num_linear_solves", "residual_evaluation_time_in_seconds",
          "num_residual_evaluations", "jacobian_evaluation_time_in_seconds", "num_jacobian_evaluations",
          "num_threads_given", "num_threads_used", "num_threads_available", "resident_memory_peak", "fraction_grouped",
          "merge_factor"]


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def convert(data):
    """What Log._convert does to a loaded log: mappings -> attribute access, lists of numbers -> arrays."""
    if isinstance(data, dict):
        return AttrDict({k: convert(v) for k, v in data.items()})
    if isinstance(data, list) and data and isinstance(data[0], Number):
        return np.array(data)
    return data


@pytest.fixture(scope="module")
def log(tmp_path_factory):
    from povar_amd import synth
    exe = os.path.join(ROOT, "build", "bal_oracle")
    if not os.path.exists(exe):
        pytest.skip("build/bal_oracle not built (python -c 'import __graft_entry__ as g; g.build()')")
    d = tmp_path_factory.mktemp("balog")
    p = synth.make_problem(10, 300, 1300, seed=21)
    f = str(d / "problem-10-300.txt")
    synth.write_data_custom(f, p)
    out = str(d / "ba_log.json")
    # a reject is wanted in the sequence: a tiny trust region makes the first trial steps fail
    r = subprocess.run([exe, "--input", f, "--log-log-path", out, "--quiet", "--max-num-iterations-step-1", "6",
                        "--max-num-iterations-step-2", "3", "--initial-trust-region-radius", "1e9"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    with open(out) as fh:
        raw = json.load(fh)
    return raw, convert(raw), p, f


def test_schema_is_the_references(log):
    raw, lg, p, path = log
    assert lg._type == "rootba_povar"
    assert set(raw) == set(PER_ITERATION) | {"_static", "_type"}
    n = len(raw["iteration"])
    for key, typ in PER_ITERATION.items():
        assert len(raw[key]) == n, key
        for v in raw[key]:
            if typ is bool:
                assert v is True or v is False, key        # JSON booleans, not 0/1
            elif typ is int:
                assert isinstance(v, int) and not isinstance(v, bool), key
            elif typ is float:
                assert isinstance(v, (int, float)) and not isinstance(v, bool), key
            else:
                assert isinstance(v, str), key
    assert list(raw) == sorted(raw)                         # nlohmann::json objects are ordered maps
    st = lg._static
    assert sorted(st) == ["problem_info", "solver", "timing"]
    assert sorted(st.problem_info) == sorted(PROBLEM_INFO) and sorted(st.timing) == sorted(TIMING)
    assert sorted(st.solver) == sorted(SOLVER)
    for s in (st.problem_info.per_lm_obs, st.problem_info.per_host_lms):
        assert sorted(s) == ["max", "mean", "min", "stddev"]


def test_values(log):
    raw, lg, p, path = log
    pi = lg._static.problem_info
    assert (pi.type, pi.input_path) == ("bal", path)
    assert (pi.num_cameras, pi.num_landmarks, pi.num_observations) == (p.n_cams, p.n_lms, p.n_obs)
    k = np.diff(p.lm_off)
    assert abs(pi.per_lm_obs.mean - k.mean()) < 1e-12 and pi.per_lm_obs.min == k.min() and pi.per_lm_obs.max == k.max()
    assert abs(pi.per_lm_obs.stddev - k.std()) < 1e-12
    # rcs_sparsity = share of empty camera-pair blocks of the reduced camera system (bal_problem.cpp:748-814)
    pairs = set()
    for l in range(p.n_lms):
        c = p.cam_idx[p.lm_off[l]:p.lm_off[l + 1]]
        pairs.update((int(a), int(b)) for a in c for b in c if b < a)
    assert abs(pi.rcs_sparsity - (1 - (p.n_cams + 2 * len(pairs)) / p.n_cams ** 2)) < 1e-12
    t = lg._static.timing
    assert t.load > 0 and t.optimize > 0 and abs(t.total - (t.load + t.preprocess + t.optimize)) < 1e-12
    sv = lg._static.solver
    assert sv.solver_type == "power_variable_projection" and sv.termination_type in ("CONVERGENCE", "NO_CONVERGENCE")
    ok = lg.step_is_successful
    # the iteration counter restarts at 0 where step 2 begins; iteration 0 counts as successful
    assert sv.num_successful_steps == int(np.sum(ok)) - 1 and sv.num_unsuccessful_steps == int(np.sum(~np.array(ok)))
    assert sv.num_threads_available >= 1 and sv.resident_memory_peak > 0 and sv.merge_factor is True
    assert np.all(lg.num_obs == p.n_obs) and np.all(lg.resident_memory > 0)
    assert np.all(np.diff(lg.cumulative_time[lg.iteration.argmin():][:3]) >= 0)
    # an unsuccessful iteration repeats the previous cost values "for monotonic plots" (ba_log_utils.cpp:124-141)
    rej = [i for i in range(1, len(ok)) if not ok[i]]
    assert rej, "the trust-region setting of the fixture is meant to produce rejected steps"
    for i in rej:
        assert lg.cost[i] == lg.cost[i - 1] and lg.cost_change[i] == 0 and lg.relative_decrease[i] == 0
    acc = [i for i in range(1, len(ok)) if ok[i] and lg.iteration[i] > 0 and ok[i - 1]]
    for i in acc:
        assert abs(lg.cost_change[i] - (lg.cost[i - 1] - lg.cost[i])) <= 1e-9 * abs(lg.cost[i - 1])
    assert abs(lg.step_solver_time[1] - (lg.scale_landmark_jacobian_time[1] + lg.perform_qr_time[1] + lg.stage2_time[1]
                                        + lg.solve_reduced_system_time[1] + lg.back_substitution_time[1])) < 1e-12
