"""GPU parity of step 2 (projective refinement, RIPOBA) against the CPU oracle and the golden
fixture.  The oracle and the HIP path use the same Householder tangent bases, so tangent-space
quantities compare directly; the golden fixture (scipy null_space bases) is compared in ambient
coordinates and basis-invariant norms only.  Tolerances as in step 1."""
import os

import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu
LAM, M = 1e-4, 10


def _state(p, seed=11):
    rng = np.random.default_rng(seed)
    cams = rng.normal(size=(p.n_cams, 12))
    cams[:, 8:11] *= 0.1
    cams[:, 11] = 5 + rng.random(p.n_cams)
    cams /= np.linalg.norm(cams, axis=1, keepdims=True)
    lms_h = np.concatenate([rng.normal(size=(p.n_lms, 3)), np.ones((p.n_lms, 1))], 1)
    return cams, lms_h, p.obs / 500.0


@pytest.mark.parametrize("e0_mode", [0, 2])
@pytest.mark.parametrize("which,norm", [("small", "NONE"), ("medium", "NONE"), ("small", "HUBER")])
def test_step2_against_oracle(which, norm, e0_mode, small_problem, medium_problem):
    from povar_amd import capi
    from oracle import povar_oracle as O
    p = small_problem if which == "small" else medium_problem
    cams, lms_h, obs = _state(p)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, obs, robust_norm=norm, huber=0.5)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, obs, robust_norm=norm, huber=0.5, e0_mode=e0_mode)
    ctx.set_cameras(cams)
    ctx.set_landmarks_homogeneous(lms_h)
    ri, ro = ctx.error_homogeneous(), orc.error_homogeneous(cams, lms_h)
    assert ri.all_num_obs == ro.all_num_obs and ri.valid_num_obs == ro.valid_num_obs
    assert abs(ri.all_error - ro.all_error) <= 1e-12 * ro.all_error
    assert abs(ri.valid_residual_sum - ro.valid_residual_sum) <= 1e-12 * ro.valid_residual_sum
    assert ctx.linearize_homogeneous()
    st_h, ok = orc.linearize_homogeneous(cams, lms_h)
    diag2 = orc.jp_diag2_homogeneous(st_h)
    jls = orc.scale_jl_cols_homogeneous(st_h)
    sigma = 1.0 / (1e-5 + np.sqrt(diag2))
    orc.scale_jp_cols_joint(st_h, sigma)
    st_n = orc.linearize_nullspace(cams, lms_h, st_h)
    hll, b, binv = orc.prepare_hb_joint(st_h, st_n, LAM)
    ref, it, status, terms = orc.solve_joint(st_n, hll, binv, b, M, want_terms=True)
    ctx.prepare_joint(LAM)
    assert rel(ctx.get_buffer(capi.BUF_DIAG2), diag2) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_JL_COL_SCALE_H), jls.ravel()) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_HLL_INV), hll.ravel()) < 1e-10
    assert rel(ctx.get_buffer(capi.BUF_B_JOINT), b) < 1e-11
    assert rel(ctx.get_buffer(capi.BUF_B_INV_JOINT), binv.ravel()) < 1e-9
    ctx.power_series_begin()
    assert rel(ctx.get_term(11), terms[0]) < 1e-11
    for i in range(1, M + 1):
        ctx.power_series_step()
        assert rel(ctx.get_term(11), terms[i]) < 1e-10, i
    assert rel(ctx.get_increment(11), ref) < 1e-10
    inc, it2, st2, rc = ctx.solve_joint(LAM, M)
    assert rc == 0 and it2 == M and rel(inc, ref) < 1e-10
    # apply_joint + the outer loop's renormalisation
    ctx.backup_joint()
    ld = ctx.apply_joint(ref)
    ld_o, lms_new = orc.back_substitute_joint(st_h, jls, LAM, cams, lms_h, ref)
    cams_new = orc.apply_cam_inc_joint(cams, ref, sigma)
    assert abs(ld - ld_o) <= 1e-9 * abs(ld_o)
    assert rel(ctx.get_cameras(), cams_new) < 1e-13 and rel(ctx.get_landmarks_homogeneous(), lms_new) < 1e-10
    ctx.normalize_joint()
    cn, ln = orc.normalize_joint(cams_new, lms_new)
    assert rel(ctx.get_cameras(), cn) < 1e-13 and rel(ctx.get_landmarks_homogeneous(), ln) < 1e-10
    ctx.restore_joint()
    assert rel(ctx.get_cameras(), cams) == 0 and rel(ctx.get_landmarks_homogeneous(), lms_h) == 0
    ctx.close()


def test_step2_against_golden_ambient():
    from povar_amd import capi
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "step2_small.npz")))
    n_c = int(g["n_cams"])
    ctx = capi.Context(n_c, g["lm_off"], g["cam_idx"], g["obs"], eps=float(g["eps"]))
    ctx.set_cameras(g["cams"])
    ctx.set_landmarks_homogeneous(g["lms_h"])
    assert abs(ctx.error_homogeneous().all_error - float(g["cost"])) <= 1e-12 * float(g["cost"])
    assert ctx.linearize_homogeneous()
    lam, m = float(g["lam"]), int(g["m"])
    ctx.prepare_joint(lam)
    assert rel(ctx.get_buffer(capi.BUF_POSE_SCALING), g["sigma"]) < 1e-13
    hw = ctx.get_buffer(capi.BUF_NC_HOUSEHOLDER).reshape(n_c, 13)

    def ambient(x):
        out = np.zeros(12 * n_c)
        for c in range(n_c):
            w, beta = hw[c, :12], hw[c, 12]
            N = (np.eye(12) - beta * np.outer(w, w))[:, 1:]
            assert np.abs(N.T @ g["cams"][c]).max() < 1e-13
            out[12 * c:12 * c + 12] = N @ x[11 * c:11 * c + 11]
        return out

    ctx.power_series_begin()
    t = ctx.get_term(11)
    assert abs(np.linalg.norm(t) / g["term_norms"][0] - 1) < 1e-11 and rel(ambient(t), g["ambient_terms"][0]) < 1e-11
    for i in range(1, m + 1):
        ctx.power_series_step()
        t = ctx.get_term(11)
        assert abs(np.linalg.norm(t) / g["term_norms"][i] - 1) < 1e-10
        assert rel(ambient(t), g["ambient_terms"][i]) < 1e-10
    inc = ctx.get_increment(11)
    assert rel(ambient(inc), g["ambient_inc"]) < 1e-10
    ld = ctx.apply_joint(inc)
    assert abs(ld - float(g["l_diff"])) <= 1e-9 * abs(float(g["l_diff"]))
    assert rel(ctx.get_cameras(), g["cams_new"]) < 1e-11 and rel(ctx.get_landmarks_homogeneous(), g["lms_new"]) < 1e-10
    ctx.normalize_joint()
    assert rel(ctx.get_cameras(), g["cams_norm"]) < 1e-11 and rel(ctx.get_landmarks_homogeneous(), g["lms_norm"]) < 1e-10
    ctx.close()


@pytest.mark.parametrize("q_tol,r_tol,m", [(1e-2, -1.0, 40), (0.0, 0.3, 40), (0.0, -1.0, 0)])
def test_step2_early_exit_and_m0(q_tol, r_tol, m, small_problem):
    """Convergence tests of solve_joint (linearization_power_varproj.hpp:255-280) and m = 0."""
    from povar_amd import capi
    from oracle import povar_oracle as O
    p = small_problem
    cams, lms_h, obs = _state(p)
    lam = 5.0
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, obs)
    st_h, ok = orc.linearize_homogeneous(cams, lms_h)
    diag2 = orc.jp_diag2_homogeneous(st_h)
    orc.scale_jl_cols_homogeneous(st_h)
    orc.scale_jp_cols_joint(st_h, 1.0 / (1e-5 + np.sqrt(diag2)))
    st_n = orc.linearize_nullspace(cams, lms_h, st_h)
    hll, b, binv = orc.prepare_hb_joint(st_h, st_n, lam)
    ref, it, status, _ = orc.solve_joint(st_n, hll, binv, b, m, q_tol=q_tol, r_tol=r_tol)
    for mode in (0, 2):
        ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, obs, e0_mode=mode)
        ctx.set_cameras(cams)
        ctx.set_landmarks_homogeneous(lms_h)
        assert ctx.linearize_homogeneous()
        inc, it2, st2, rc = ctx.solve_joint(lam, m, q_tol, r_tol)
        assert rc == 0 and (it2, st2) == (it, status)
        assert rel(inc, ref) < 1e-10
        ctx.close()
