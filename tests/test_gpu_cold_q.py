"""Cold observations of the lane-per-landmark kernels (camera not in the workgroup's LDS set): where they leave their
scatter scalars q.  Two addressings of Dp::q4c, chosen by the share of cold observations (from 20 % on; POVAR_COLD_Q_ROWS overrides):
straight into the camera-major cold view (one scattered 32-byte store per lane) or row-major next to the other lanes of
the row (lpl_cold_q), gathered by the per-camera kernels through CmView::src.  Both against the oracle, term by term,
steps 1 and 2, b of prepare_Hb included, with camera sets of 4 / 8 slots so that most observations are cold."""
import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu
ALPHA, LAM, M = 0.01, 1e-4, 8


def _env(monkeypatch, q_rows, hot_acc):
    monkeypatch.setenv("POVAR_E0_V1", "0")          # lane-per-landmark kernels also under 65 536 observations
    monkeypatch.setenv("POVAR_HOT_ACC", str(hot_acc))
    monkeypatch.setenv("POVAR_COLD_Q_ROWS", q_rows)


@pytest.mark.parametrize("hot_acc", [4, 8])
@pytest.mark.parametrize("q_rows", ["0", "1"])
@pytest.mark.parametrize("norm", ["NONE", "HUBER"])
def test_step1_cold_q(norm, q_rows, hot_acc, medium_problem, monkeypatch):
    from povar_amd import capi
    from oracle import povar_oracle as O
    p = medium_problem
    _env(monkeypatch, q_rows, hot_acc)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=norm, huber=3.0)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=norm, huber=3.0, e0_mode=capi.E0_IMPLICIT_LDSACC)
    li = ctx.layout_info()
    assert li.lane_per_landmark == 1 and li.n_cold > 0.2 * li.n_obs
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    assert ctx.linearize_pose(ALPHA)
    st, diag2, jls, sigma, ok = orc.stage1_pose(ALPHA, p.cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, LAM, 0.0)
    ref, it, status, terms = orc.solve_pose(st, hll, binv, b, M, want_terms=True)
    ctx.prepare_pose(LAM)
    assert rel(ctx.get_buffer(capi.BUF_B), b) < 1e-11            # prepare_lpl's cold observations go the same way
    ctx.power_series_begin()
    for i in range(1, M + 1):
        ctx.power_series_step()
        assert rel(ctx.get_term(), terms[i]) < 1e-11, i
    inc, it_g, st_g, rc = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)  # the hipGraph of the whole loop
    assert rc == 0 and rel(inc, ref) < 1e-10
    ctx.close()


@pytest.mark.parametrize("hot_acc", [4, 8])
@pytest.mark.parametrize("q_rows", ["0", "1"])
def test_step2_cold_q(q_rows, hot_acc, medium_problem, monkeypatch):
    from povar_amd import capi
    from oracle import povar_oracle as O
    from test_gpu_step2 import _state
    p = medium_problem
    _env(monkeypatch, q_rows, hot_acc)
    cams, lms_h, obs = _state(p)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, obs)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    assert ctx.layout_info().n_cold > 0.2 * p.n_obs
    ctx.set_cameras(cams)
    ctx.set_landmarks_homogeneous(lms_h)
    assert ctx.linearize_homogeneous()
    st_h, ok = orc.linearize_homogeneous(cams, lms_h)
    diag2 = orc.jp_diag2_homogeneous(st_h)
    orc.scale_jl_cols_homogeneous(st_h)
    orc.scale_jp_cols_joint(st_h, 1.0 / (1e-5 + np.sqrt(diag2)))
    st_n = orc.linearize_nullspace(cams, lms_h, st_h)
    hll, b, binv = orc.prepare_hb_joint(st_h, st_n, LAM)
    ref, it, status, terms = orc.solve_joint(st_n, hll, binv, b, M, want_terms=True)
    ctx.prepare_joint(LAM)
    assert rel(ctx.get_buffer(capi.BUF_B_JOINT), b) < 1e-11
    ctx.power_series_begin()
    for i in range(1, M + 1):
        ctx.power_series_step()
        assert rel(ctx.get_term(11), terms[i]) < 1e-10, i
    inc, it2, st2, rc = ctx.solve_joint(LAM, M)
    assert rc == 0 and rel(inc, ref) < 1e-10
    ctx.close()
