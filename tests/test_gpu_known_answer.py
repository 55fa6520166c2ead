"""GPU half of tests/test_known_answer.py: the same known-answer configurations through `bin/bal` (host LM loops over
the C ABI of libpovar_hip.so).  A converged run must end at the chi-square cost of the generator's 0.5 px noise --
a check against geometry, not against the oracle.  The small problems run with the kernels the library would pick
for them (lane per observation) and with the headline lane-per-landmark kernels forced (POVAR_E0_V1=0)."""
import os

import pytest

from test_known_answer import COMMON, KNOWN_ANSWER, check_floor, run_bal, write_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("family", ["auto", "lane-per-landmark", "resident-series"])
@pytest.mark.parametrize("shape,seed,flags", KNOWN_ANSWER)
def test_bal_hip_reaches_noise_floor(tmp_path, monkeypatch, shape, seed, flags, family):
    if family in ("lane-per-landmark", "resident-series"):
        if "CHOLESKY" in flags and "RIPCG" in flags:
            pytest.skip("no power-series kernel on this route")
    if family == "resident-series":  # step 1's power series as ONE launch (series_res) in every LM iteration of the run
        monkeypatch.setenv("POVAR_RES", "1")
        monkeypatch.delenv("POVAR_E0_V1", raising=False)
    elif family == "lane-per-landmark":
        monkeypatch.setenv("POVAR_E0_V1", "0")
    else:
        monkeypatch.delenv("POVAR_E0_V1", raising=False)
    p, f = write_problem(tmp_path, shape, seed)
    res = run_bal("bin/bal", f, str(tmp_path / "log.json"), flags + COMMON + ["--quiet"])
    check_floor(p, res)


def test_bal_hip_truncation_stall_and_cure(tmp_path):
    """Same statement as test_random_start_stall_is_the_route_not_the_algebra, on the HIP path."""
    p, f = write_problem(tmp_path, (10, 300, 1300), 21)
    common = ["--solver-type-step-2", "RIPOBA", "--max-num-iterations-step-1", "100", "--max-num-iterations-step-2", "300", "--quiet"]
    short = run_bal("bin/bal", f, str(tmp_path / "a.json"), ["--solver-type-step-1", "POWER_VARPROJ", "--power-sc-iterations", "20"] + common)
    long = run_bal("bin/bal", f, str(tmp_path / "b.json"),
                   ["--solver-type-step-1", "POWER_VARPROJ", "--power-sc-iterations", "500", "--eta", "0"] + common)
    assert short["final"][-1][1] > 5.0
    check_floor(p, long)
