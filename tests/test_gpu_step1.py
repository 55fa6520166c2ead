"""GPU parity of the step-1 (pOSE / VarPro) hot path against the CPU oracle, through the C ABI.

Tolerances: per power-series term 1e-12 relative, after 20 terms 1e-10;
fp64 throughout.  K1 (landmark init) uses 3x3 normal equations on the GPU where the reference
uses an SVD solve: tolerance 1e-8.
"""
import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu

ALPHA, LAM, M = 0.01, 1e-4, 20


def _setup(p, norm="NONE", huber=1.0, e0_mode=0):
    from povar_amd import capi
    from oracle import povar_oracle as O
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=norm, huber=huber)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=norm, huber=huber, e0_mode=e0_mode)
    return orc, ctx


def _oracle_stage(orc, cams, lms, lam, lam_lm=0.0):
    st, diag2, jls, sigma, ok = orc.stage1_pose(ALPHA, cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, lam, lam_lm)
    return st, diag2, jls, sigma, hll, b, binv


@pytest.mark.parametrize("which", ["small", "medium"])
def test_init_and_error(which, small_problem, medium_problem):
    p = small_problem if which == "small" else medium_problem
    orc, ctx = _setup(p)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    assert rel(ctx.get_landmarks(), lms) < 1e-11
    # identical landmarks from here on
    ctx.set_landmarks(lms)
    ri, ro = ctx.error_pose(ALPHA), orc.error_pose(ALPHA, p.cams, lms)
    assert ri.all_num_obs == ro.all_num_obs == p.n_obs
    assert abs(ri.all_error - ro.all_error) <= 1e-12 * ro.all_error
    assert abs(ri.all_residual_sum - ro.all_residual_sum) <= 1e-12 * ro.all_residual_sum
    assert ri.is_numerically_valid == 1 and ri.valid_num_obs == ro.valid_num_obs
    ctx.close()


@pytest.mark.parametrize("norm", ["NONE", "HUBER", "CAUCHY"])
def test_linearize_prepare_buffers(norm, small_problem):
    from povar_amd import capi
    p = small_problem
    orc, ctx = _setup(p, norm, huber=30.0)
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    assert ctx.linearize_pose(ALPHA)
    st, diag2, jls, sigma, hll, b, binv = _oracle_stage(orc, p.cams, lms, LAM)
    ctx.prepare_pose(LAM)
    assert rel(ctx.get_buffer(capi.BUF_DIAG2), diag2) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_POSE_SCALING), sigma) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_JL_COL_SCALE), jls.ravel()) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_STORAGE), st.ravel()) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_HLL_INV), hll.ravel()) < 1e-11
    assert rel(ctx.get_buffer(capi.BUF_B), b) < 1e-12
    assert rel(ctx.get_buffer(capi.BUF_B_INV), binv.ravel()) < 1e-10
    ctx.close()


@pytest.mark.parametrize("force_lpl", ["0", None])
def test_exports_belong_to_the_linearisation_alpha(force_lpl, small_problem, monkeypatch):
    """The lazily rebuilt per-slot arrays behind the exports (and the POBA back-substitution, the SC solvers) are built
    with the alpha of povar_linearize_pose, also when a cost evaluation at ANOTHER alpha ran in between (ADVICE r02:
    ensure_legacy used whatever alpha the last call left in the context)."""
    from povar_amd import capi
    if force_lpl is not None:
        monkeypatch.setenv("POVAR_E0_V1", force_lpl)   # the small problem through the lane-per-landmark kernels: lazy arrays
    p = small_problem
    orc, ctx = _setup(p, "HUBER", huber=30.0, e0_mode=capi.E0_IMPLICIT_LDSACC)
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    assert ctx.linearize_pose(ALPHA)
    st, diag2, jls, sigma, hll, b, binv = _oracle_stage(orc, p.cams, lms, LAM)
    ctx.prepare_pose(LAM)
    ctx.error_pose(0.37)                                 # leaves another alpha in the context
    assert rel(ctx.get_buffer(capi.BUF_STORAGE), st.ravel()) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_JL_COL_SCALE), jls.ravel()) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_HLL_INV), hll.ravel()) < 1e-11
    ctx.close()


@pytest.mark.parametrize("e0_mode", [0, 1, 2, 3])
@pytest.mark.parametrize("which", ["small", "medium"])
def test_power_series_term_by_term(which, e0_mode, small_problem, medium_problem):
    p = small_problem if which == "small" else medium_problem
    orc, ctx = _setup(p, e0_mode=e0_mode)
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    assert ctx.linearize_pose(ALPHA)
    st, diag2, jls, sigma, hll, b, binv = _oracle_stage(orc, p.cams, lms, LAM)
    ref, it, status, terms = orc.solve_pose(st, hll, binv, b, M, want_terms=True)
    ctx.prepare_pose(LAM)
    x = np.random.default_rng(0).normal(size=12 * p.n_cams)
    assert rel(ctx.right_mul_e0_pose(x), orc.right_mul_e0_pose(st, hll, x)) < 1e-12
    ctx.power_series_begin()
    assert rel(ctx.get_term(), terms[0]) < 1e-12
    for i in range(1, M + 1):
        ctx.power_series_step()
        assert rel(ctx.get_term(), terms[i]) < 1e-11, i
    assert rel(ctx.get_increment(), ref) < 1e-10
    # one-shot entry point
    inc, it2, st2, rc = ctx.solve_pose(LAM, 0, M)
    assert rc == 0 and it2 == M and st2 == 0 and rel(inc, ref) < 1e-10
    ctx.close()


@pytest.mark.parametrize("e0_mode", [0, 2, 3])  # 2, 3: norm partials come from the fused cam_cold_sum_binv
@pytest.mark.parametrize("q_tol,r_tol", [(1e-2, -1.0), (0.0, 0.5), (0.3, 0.9)])
def test_early_exit(q_tol, r_tol, e0_mode, small_problem):
    p = small_problem
    orc, ctx = _setup(p, e0_mode=e0_mode)
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    ctx.linearize_pose(ALPHA)
    lam = 10.0  # strong damping so the series actually converges inside m terms
    st, diag2, jls, sigma, hll, b, binv = _oracle_stage(orc, p.cams, lms, lam)
    ref, it, status, _ = orc.solve_pose(st, hll, binv, b, 50, q_tol=q_tol, r_tol=r_tol)
    inc, it2, st2, rc = ctx.solve_pose(lam, 0, 50, q_tol, r_tol)
    assert (it2, st2) == (it, status)
    assert rel(inc, ref) < 1e-10
    ctx.close()


@pytest.mark.parametrize("solver", [0, 1])
def test_apply(solver, medium_problem):
    p = medium_problem
    orc, ctx = _setup(p)
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    ctx.linearize_pose(ALPHA)
    lam_lm = LAM if solver == 1 else 0.0
    st, diag2, jls, sigma, hll, b, binv = _oracle_stage(orc, p.cams, lms, LAM, lam_lm)
    ref, _, _, _ = orc.solve_pose(st, hll, binv, b, M)
    inc, _, _, rc = ctx.solve_pose(LAM, solver, M)
    assert rc == 0 and rel(inc, ref) < 1e-10
    ctx.backup_pose()
    l_diff = ctx.apply_pose(solver, ALPHA, ref)
    inc_s = ref * sigma
    cams_new = p.cams + inc_s.reshape(-1, 12)
    if solver == 0:
        ld, lms_new = orc.back_substitute_pose(ALPHA, st, cams_new, lms, inc_s * (1.0 / sigma))
    else:
        ld, lms_new = orc.back_substitute_poba(st, jls, LAM, lms, ref)
    assert rel(ctx.get_cameras(), cams_new) < 1e-14
    assert rel(ctx.get_landmarks(), lms_new) < 1e-9
    assert abs(l_diff - ld) <= 1e-9 * abs(ld)
    ctx.restore_pose()
    assert rel(ctx.get_cameras(), p.cams) == 0 and rel(ctx.get_landmarks(), lms) == 0
    ctx.close()


def test_long_landmarks():
    """Landmarks with more than 64 observations take the lm_long driver."""
    from povar_amd import synth
    rng = np.random.default_rng(5)
    n_c = 150
    ks = [2, 130, 3, 64, 65, 5, 100, 2, 7, 150]
    lm_off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    cam_idx = np.concatenate([np.sort(rng.choice(n_c, k, replace=False)) for k in ks]).astype(np.int32)
    base = synth.make_problem(n_c, 400, 1700, seed=1)
    obs = rng.normal(scale=100.0, size=(cam_idx.shape[0], 2))
    from povar_amd import capi
    from oracle import povar_oracle as O
    for e0_mode in (0, 1, 2, 3):
        orc = O.Oracle(n_c, lm_off, cam_idx, obs)
        ctx = capi.Context(n_c, lm_off, cam_idx, obs, e0_mode=e0_mode)
        lms = orc.init_landmarks_pose(ALPHA, base.cams)
        ctx.set_cameras(base.cams)
        ctx.init_landmarks_pose(ALPHA)
        assert rel(ctx.get_landmarks(), lms) < 1e-11
        ctx.set_landmarks(lms)
        ctx.linearize_pose(ALPHA)
        st, diag2, jls, sigma, hll, b, binv = _oracle_stage(orc, base.cams, lms, LAM)
        ref, _, _, _ = orc.solve_pose(st, hll, binv, b, 10)
        inc, it, stt, rc = ctx.solve_pose(LAM, 0, 10)
        assert rc == 0 and rel(inc, ref) < 1e-10
        l_diff = ctx.apply_pose(0, ALPHA, ref)
        inc_s = ref * sigma
        ld, lms_new = orc.back_substitute_pose(ALPHA, st, base.cams + inc_s.reshape(-1, 12), lms, inc_s * (1.0 / sigma))
        assert rel(ctx.get_landmarks(), lms_new) < 1e-9 and abs(l_diff - ld) <= 1e-9 * abs(ld)
        ctx.close()


@pytest.mark.parametrize("name,norm", [("step1_small_none.npz", "NONE"), ("step1_small_huber.npz", "HUBER"),
                                       ("step1_medium_none.npz", "NONE")])
def test_against_golden_fixtures(name, norm):
    """HIP path vs the committed golden vectors (tests/golden, NumPy restatement)."""
    import os
    from povar_amd import capi
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", name)))
    ctx = capi.Context(int(g["n_cams"]), g["lm_off"], g["cam_idx"], g["obs"], robust_norm=norm,
                       huber=float(g["huber"]), eps=float(g["eps"]))
    alpha, lam, m = float(g["alpha"]), float(g["lam"]), int(g["m"])
    ctx.set_cameras(g["cams"])
    ctx.init_landmarks_pose(alpha)
    assert rel(ctx.get_landmarks(), g["lms"]) < 1e-11
    ctx.set_landmarks(g["lms"])
    ri = ctx.error_pose(alpha)
    assert abs(ri.all_error - float(g["cost"])) <= 1e-12 * float(g["cost"])
    assert ctx.linearize_pose(alpha)
    ctx.prepare_pose(lam)
    assert rel(ctx.get_buffer(capi.BUF_DIAG2), g["diag2"]) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_JL_COL_SCALE), g["jl_scale"].ravel()) < 1e-13
    assert rel(ctx.get_buffer(capi.BUF_HLL_INV), g["hll_inv"].ravel()) < 1e-11
    assert rel(ctx.get_buffer(capi.BUF_B), g["b"]) < 1e-12
    if "storage" in g:
        assert rel(ctx.get_buffer(capi.BUF_STORAGE), g["storage"].ravel()) < 1e-13
    ctx.power_series_begin()
    assert rel(ctx.get_term(), g["terms"][0]) < 1e-12
    for i in range(1, m + 1):
        ctx.power_series_step()
        assert rel(ctx.get_term(), g["terms"][i]) < 1e-11, i
    assert rel(ctx.get_increment(), g["inc"]) < 1e-10
    if "varproj_l_diff" in g:
        ld = ctx.apply_pose(0, alpha, g["inc"])
        assert rel(ctx.get_cameras(), g["varproj_cams_new"]) < 1e-14
        assert rel(ctx.get_landmarks(), g["varproj_lms_new"]) < 1e-9
        assert abs(ld - float(g["varproj_l_diff"])) <= 1e-9 * abs(float(g["varproj_l_diff"]))
    ctx.close()


def test_full_size_properties():
    """Size-independent properties at a BASELINE size (trafalgar-257 shape): E0 symmetric PSD,
    stored-tile and implicit operators agree, determinism run to run, observation count."""
    from povar_amd import capi, synth
    p = synth.make_bal_problem("trafalgar-257")
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    assert ctx.error_pose(ALPHA).all_num_obs == p.n_obs
    assert ctx.linearize_pose(ALPHA)
    ctx.prepare_pose(LAM)
    rng = np.random.default_rng(1)
    x, y = rng.normal(size=12 * p.n_cams), rng.normal(size=12 * p.n_cams)
    ex, ey = ctx.right_mul_e0_pose(x), ctx.right_mul_e0_pose(y)
    assert abs(y @ ex - x @ ey) <= 1e-11 * abs(y @ ex)
    assert x @ ex > 0
    assert np.array_equal(ctx.right_mul_e0_pose(x), ex)  # deterministic (no atomics on the path)
    inc_a, it, st, rc = ctx.solve_pose(LAM, 0, M)
    for mode in (capi.E0_TILES, capi.E0_IMPLICIT_LDSACC, capi.E0_TILES_LDSACC):
        ctx.set_e0_mode(mode)
        assert rel(ctx.right_mul_e0_pose(x), ex) < 1e-12  # per-term bar (the forms differ in summation order only)
        inc_b, _, _, _ = ctx.solve_pose(LAM, 0, M)
        assert rc == 0 and rel(inc_b, inc_a) < 1e-11
    # S = B - E0 is positive definite: x^T B x > x^T E0 x with B^-1 from the library
    binv = ctx.get_buffer(capi.BUF_B_INV).reshape(p.n_cams, 12, 12)
    xb = x.reshape(p.n_cams, 12)
    xBx = sum(v @ np.linalg.solve(bi, v) for v, bi in zip(xb, binv))
    assert xBx > x @ ex
    ctx.close()


def test_venice_size_properties():
    """BASELINE.json's headline size (venice-1778 shape, 5M observations): size-independent
    properties of the operator -- symmetry, positivity, agreement of the two E0 forms, bit
    reproducibility -- and the robust (HUBER) path at scale."""
    from povar_amd import capi, synth
    p = synth.make_bal_problem("venice-1778")
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm="HUBER", huber=50.0)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    ri = ctx.error_pose(ALPHA)
    assert ri.all_num_obs == p.n_obs and ri.is_numerically_valid == 1
    assert ctx.linearize_pose(ALPHA)
    ctx.prepare_pose(LAM)
    rng = np.random.default_rng(2)
    x, y = rng.normal(size=12 * p.n_cams), rng.normal(size=12 * p.n_cams)
    ex, ey = ctx.right_mul_e0_pose(x), ctx.right_mul_e0_pose(y)
    assert abs(y @ ex - x @ ey) <= 1e-10 * abs(y @ ex) and x @ ex > 0
    assert np.array_equal(ctx.right_mul_e0_pose(x), ex)
    inc_a, it, st, rc = ctx.solve_pose(LAM, 0, M)
    assert rc == 0 and it == M
    for mode in (capi.E0_TILES, capi.E0_IMPLICIT_LDSACC, capi.E0_TILES_LDSACC):
        ctx.set_e0_mode(mode)
        assert rel(ctx.right_mul_e0_pose(x), ex) < 1e-12
        inc_b, _, _, _ = ctx.solve_pose(LAM, 0, M)
        assert rel(inc_b, inc_a) < 1e-11
    # accepted-step property of the model: apply, then the cost must be finite and the state moves
    ctx.set_e0_mode(capi.E0_IMPLICIT)
    ctx.backup_pose()
    l_diff = ctx.apply_pose(0, ALPHA, inc_a)
    ri2 = ctx.error_pose(ALPHA)
    assert np.isfinite(l_diff) and ri2.is_numerically_valid == 1 and ri2.all_num_obs == p.n_obs
    ctx.restore_pose()
    assert abs(ctx.error_pose(ALPHA).all_error - ri.all_error) == 0
    ctx.close()


def test_unobserved_cameras_and_two_view_landmarks():
    """Edge cases of the layout: cameras that no landmark observes (empty camera-major segments,
    sigma = 1/eps, B = lambda I) and a problem made of 2-observation landmarks only."""
    from povar_amd import capi
    from oracle import povar_oracle as O
    rng = np.random.default_rng(9)
    n_c, n_l = 12, 200
    used = np.array([0, 2, 3, 5, 7, 8, 11])                      # cameras 1, 4, 6, 9, 10 are never observed
    cam_idx = np.concatenate([np.sort(rng.choice(used, 2, replace=False)) for _ in range(n_l)]).astype(np.int32)
    lm_off = (2 * np.arange(n_l + 1)).astype(np.int32)
    obs = rng.normal(scale=50.0, size=(2 * n_l, 2))
    cams = np.zeros((n_c, 12))
    cams[:, :8] = rng.normal(size=(n_c, 8))
    cams[:, 11] = 1.0
    orc = O.Oracle(n_c, lm_off, cam_idx, obs)
    for mode in (0, 1, 2, 3):
        ctx = capi.Context(n_c, lm_off, cam_idx, obs, e0_mode=mode)
        ctx.set_cameras(cams)
        ctx.init_landmarks_pose(ALPHA)
        lms = orc.init_landmarks_pose(ALPHA, cams)
        assert rel(ctx.get_landmarks(), lms) < 1e-10
        ctx.set_landmarks(lms)
        assert ctx.linearize_pose(ALPHA)
        st, diag2, jls, sigma, hll, b, binv = _oracle_stage(orc, cams, lms, LAM)
        ref, _, _, _ = orc.solve_pose(st, hll, binv, b, 10)
        inc, it, stt, rc = ctx.solve_pose(LAM, 0, 10)
        assert rc == 0 and rel(inc, ref) < 1e-10
        assert rel(ctx.get_buffer(capi.BUF_POSE_SCALING), sigma) < 1e-13
        unused = np.setdiff1d(np.arange(n_c), used)
        assert np.all(inc.reshape(n_c, 12)[unused] == 0)         # b = 0 there, so the increment is exactly 0
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_landmark_order_invariance(seed, medium_problem):
    """the same problem with its landmarks in another order (different wave bins, different
    camera-major item cuts, different hot/cold split inside the bins) gives the same per-camera results
    to reduction tolerance, in every E0 mode, for the power series and for PCG."""
    from povar_amd import capi
    p = medium_problem
    perm = np.random.default_rng(seed).permutation(p.n_lms)
    k = np.diff(p.lm_off)
    lm_off2 = np.concatenate([[0], np.cumsum(k[perm])]).astype(np.int32)
    idx = np.concatenate([np.arange(p.lm_off[l], p.lm_off[l + 1]) for l in perm])

    def run(lm_off, cam_idx, obs, mode):
        ctx = capi.Context(p.n_cams, lm_off, cam_idx, obs, e0_mode=mode)
        ctx.set_cameras(p.cams)
        ctx.init_landmarks_pose(ALPHA)
        lms = ctx.get_landmarks()
        cost = ctx.error_pose(ALPHA).all_error
        assert ctx.linearize_pose(ALPHA)
        inc, it, st, rc = ctx.solve_pose(LAM, 0, M)
        pcg, it_p, st_p, rc_p = ctx.solve_pose_sc(LAM, capi.SC_PCG, 0, 500, 1e-2)
        ld = ctx.apply_pose(0, ALPHA, inc)
        out = dict(lms=lms, cost=cost, inc=inc, pcg=pcg, it_p=it_p, ld=ld, lms_new=ctx.get_landmarks(),
                   sigma=ctx.get_buffer(capi.BUF_POSE_SCALING))
        ctx.close()
        return out

    for mode in (0, 1, 2, 3):
        a = run(p.lm_off, p.cam_idx, p.obs, mode)
        b = run(lm_off2, p.cam_idx[idx], p.obs[idx], mode)
        # per-landmark work does not see the order, except for the association of the wavefront scan
        # (it depends on where the 16-lane DPP rows cut a landmark's segment)
        assert rel(b["lms"], a["lms"][perm]) < 1e-13
        assert abs(a["cost"] - b["cost"]) <= 1e-12 * a["cost"]
        assert rel(b["sigma"], a["sigma"]) < 1e-13 and rel(b["inc"], a["inc"]) < 1e-10
        assert b["it_p"] == a["it_p"] and rel(b["pcg"], a["pcg"]) < 1e-9
        assert abs(a["ld"] - b["ld"]) <= 1e-9 * abs(a["ld"]) and rel(b["lms_new"], a["lms_new"][perm]) < 1e-9


@pytest.mark.parametrize("e0_mode", [0, 2])
def test_e0_and_b_against_exact_rational_arithmetic(e0_mode, _term_kernels):
    """the HIP path against EXACT rational arithmetic evaluated straight from the reference's formulas
    (tests/exact_rational.py: fractions.Fraction on the input doubles; bal_bundle_adjustment_helper.cpp:251-310,
    landmark_block.hpp:517-536, linearization_power_varproj.hpp:377-396) on the 6-camera golden problem -- independent of the C
    oracle and of the NumPy restatement.  E0_scaled x = sigma * E0 (sigma * x), b_scaled = sigma * b with the library's OWN
    pose scaling taken as exact numbers (the scaling itself against a 60-digit evaluation): 1e-13, in every kernel family the
    module runs in."""
    import os
    from fractions import Fraction as F
    from povar_amd import capi
    from exact_rational import ExactStep1, rel_err, sigma_60_digits
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step1_small_none.npz"))
    n_c = int(g["n_cams"])
    ex = ExactStep1(float(g["alpha"]), n_c, g["lm_off"], g["cam_idx"], g["obs"], g["cams"], g["lms"])
    ctx = capi.Context(n_c, g["lm_off"], g["cam_idx"], g["obs"], eps=float(g["eps"]), e0_mode=e0_mode)
    ctx.set_cameras(g["cams"])
    ctx.set_landmarks(g["lms"])
    assert ctx.linearize_pose(float(g["alpha"]))
    ctx.prepare_pose(float(g["lam"]))
    d = ex.diag2()
    assert rel_err(ctx.get_buffer(capi.BUF_DIAG2), d) < 1e-14
    sigma = ctx.get_buffer(capi.BUF_POSE_SCALING)
    assert np.abs(sigma / np.array(sigma_60_digits(d, float(g["eps"]))) - 1).max() < 1e-14
    sg = [F(float(t)) for t in sigma]
    assert rel_err(ctx.get_buffer(capi.BUF_B), [s * t for s, t in zip(sg, ex.b())]) < 1e-13
    x = np.random.default_rng(5).normal(size=12 * n_c)
    y_exact = [s * t for s, t in zip(sg, ex.e0([F(float(a)) * s for a, s in zip(x, sg)]))]
    y = ctx.right_mul_e0_pose(x)
    assert rel_err(y, y_exact) < 1e-13, rel_err(y, y_exact)
    ctx.close()
