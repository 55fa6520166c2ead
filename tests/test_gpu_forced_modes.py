"""The at-size oracle comparison of step 1 under FORCED library modes -- paths the defaults take only for other graphs or
sizes.  tools/forced_mode_suite.sh runs the whole GPU suite under nine such modes by hand; these are in `-m gpu` so that
every run of the suite (the driver's included) covers them at the trafalgar-257 and venice-1778 sizes:

  * the range strategy (contiguous landmark ranges, per-workgroup camera sets) on the graph WITHOUT locality, with only 8
    LDS camera slots: most observations are cold, the cold view is large, q leaves the row kernels row-major;
  * rows placed on a host thread and swapped in later (POVAR_LPL_PLACE=async) with the row-major cold q forced on;
  * the camera-chunk E0 kernel forced (the automatic choice may or may not take it on a given box).
"""
import pytest

pytestmark = pytest.mark.gpu

MODES = {
    "range-strategy-8-slots": {"POVAR_E0_V1": "0", "POVAR_LPL_STRATEGY": "range", "POVAR_HOT_ACC": "8"},
    "async-placement-row-major-cold-q": {"POVAR_E0_V1": "0", "POVAR_LPL_PLACE": "async", "POVAR_COLD_Q_ROWS": "1"},
    "camera-chunk-kernel": {"POVAR_E0_V1": "0", "POVAR_E0_CK": "1", "POVAR_LPL_PLACE": "sync"},
}


@pytest.mark.parametrize("name", ["trafalgar-257", "venice-1778"])
@pytest.mark.parametrize("mode", sorted(MODES))
def test_step1_oracle_parity_under_forced_modes(mode, name, monkeypatch):
    from test_gpu_baseline_sizes import step1_oracle_parity
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    step1_oracle_parity(name)
