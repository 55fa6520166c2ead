"""Known-answer / self-consistency tests of the oracle (the reference has no tests of its own;
its helper's method -- central differences -- is reused: testing/test_jacobian.hpp:49-107)."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel


def _lin_pose(alpha, obs, x, P):
    from oracle import povar_oracle as O
    res, Jp, Jl = np.zeros(4), np.zeros((4, 12)), np.zeros((4, 3))
    O.lib().orc_linearize_point_pose(C.c_double(alpha), C.c_void_p(obs.ctypes.data), C.c_void_p(x.ctypes.data),
                                     C.c_void_p(P.ctypes.data), C.c_void_p(res.ctypes.data),
                                     C.c_void_p(Jp.ctypes.data), C.c_void_p(Jl.ctypes.data))
    return res, Jp, Jl


def _lin_hom(obs, X, P):
    from oracle import povar_oracle as O
    res, Jp, Jl = np.zeros(2), np.zeros((2, 12)), np.zeros((2, 4))
    O.lib().orc_linearize_point_homogeneous(C.c_void_p(obs.ctypes.data), C.c_void_p(X.ctypes.data),
                                            C.c_void_p(P.ctypes.data), C.c_void_p(res.ctypes.data),
                                            C.c_void_p(Jp.ctypes.data), C.c_void_p(Jl.ctypes.data))
    return res, Jp, Jl


def _numeric(f, x0, eps=1e-6):
    cols = []
    for k in range(x0.size):
        d = np.zeros_like(x0)
        d[k] = eps
        cols.append((f(x0 + d) - f(x0 - d)) / (2 * eps))
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("seed", range(5))
def test_jacobians_pose_central_differences(seed):
    rng = np.random.default_rng(seed)
    alpha, obs, x, P = 0.01 + 0.2 * rng.random(), rng.normal(size=2) * 50, rng.normal(size=3), rng.normal(size=12)
    res, Jp, Jl = _lin_pose(alpha, obs, x, P)
    Jp_n = _numeric(lambda q: _lin_pose(alpha, obs, x, q)[0], P)
    Jl_n = _numeric(lambda q: _lin_pose(alpha, obs, q, P)[0], x)
    assert rel(Jp, Jp_n) < 1e-7 and rel(Jl, Jl_n) < 1e-7


@pytest.mark.parametrize("seed", range(5))
def test_jacobians_homogeneous_central_differences(seed):
    rng = np.random.default_rng(100 + seed)
    obs, X, P = rng.normal(size=2), np.append(rng.normal(size=3), 1.0), rng.normal(size=12)
    P[11] += 4.0
    res, Jp, Jl = _lin_hom(obs, X, P)
    assert rel(Jp, _numeric(lambda q: _lin_hom(obs, X, q)[0], P)) < 1e-6
    assert rel(Jl, _numeric(lambda q: _lin_hom(obs, q, P)[0], X)) < 1e-6


def test_error_weight_kinds():
    from oracle import povar_oracle as O
    L = O.lib()
    e, w = C.c_double(), C.c_double()
    for norm, r2, exp_e, exp_w in [(0, 4.0, 2.0, 1.0), (1, 0.25, 0.125, 1.0), (1, 16.0, 0.5 * (2 - 0.25) * 0.25 * 16, 0.25),
                                   (2, 3.0, np.log(4.0), 1.0)]:
        o = O._Options(norm, 1.0, 1e-5)
        L.orc_error_weight(C.byref(o), C.c_double(r2), C.byref(e), C.byref(w))
        assert abs(e.value - exp_e) < 1e-15 and abs(w.value - exp_w) < 1e-15


def test_series_converges_to_schur_solve(small_problem):
    """m -> infinity with strong damping: the power series equals the dense S^-1(-b)."""
    from oracle import povar_oracle as O, povar_numpy as N
    p = small_problem
    alpha, lam = 0.01, 50.0
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    lms = orc.init_landmarks_pose(alpha, p.cams)
    st, diag2, jls, sigma, ok = orc.stage1_pose(alpha, p.cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, lam)
    s1 = N.step1(alpha, p.n_cams, p.lm_off, p.cam_idx, p.obs, p.cams, lms, lam, 1)
    assert s1["rho"] < 0.9
    inc, it, status, _ = orc.solve_pose(st, hll, binv, b, 400)
    assert rel(inc, s1["exact"]) < 1e-10
    # E0 is symmetric positive semi-definite
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=12 * p.n_cams), rng.normal(size=12 * p.n_cams)
    assert abs(x @ orc.right_mul_e0_pose(st, hll, y) - y @ orc.right_mul_e0_pose(st, hll, x)) < 1e-9 * abs(x @ orc.right_mul_e0_pose(st, hll, y))
    assert x @ orc.right_mul_e0_pose(st, hll, x) >= 0
    assert rel(orc.right_mul_e0_pose(st, hll, x), s1["E0"] @ x) < 1e-12
    # threaded (per-camera mutex) variant agrees with the serial one
    assert rel(orc.right_mul_e0_pose(st, hll, x, n_threads=4), orc.right_mul_e0_pose(st, hll, x)) < 1e-14


def test_early_exit_semantics(small_problem):
    """q_tolerance / r_tolerance behaviour of solve_pOSE (linearization_power_varproj.hpp:206-229)."""
    from oracle import povar_oracle as O
    p = small_problem
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    lms = orc.init_landmarks_pose(0.01, p.cams)
    st, diag2, jls, sigma, ok = orc.stage1_pose(0.01, p.cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, 10.0)
    inc, it, status, terms = orc.solve_pose(st, hll, binv, b, 50, q_tol=1e-2, want_terms=True)
    assert status == 1 and 0 < it < 50
    acc = np.cumsum(terms[: it + 1], axis=0)
    zeta = [i * np.linalg.norm(terms[i]) / np.linalg.norm(acc[i]) for i in range(1, it + 1)]
    assert zeta[-1] < 1e-2 and all(z >= 1e-2 for z in zeta[:-1])
    # m == 0: block-Jacobi step only
    inc0, it0, status0, _ = orc.solve_pose(st, hll, binv, b, 0)
    assert it0 == 0 and status0 == 0 and rel(inc0, orc.right_mul_b_inv(binv, -b)) == 0
    # tolerances <= 0 disable the tests
    inc_m, it_m, status_m, _ = orc.solve_pose(st, hll, binv, b, 7, q_tol=0.0, r_tol=-1.0)
    assert it_m == 7 and status_m == 0


def test_kernel_basis_orthonormal():
    from oracle import povar_oracle as O
    rng = np.random.default_rng(3)
    for n in (4, 12):
        for _ in range(5):
            v = rng.normal(size=n)
            Nn = np.zeros((n, n - 1))
            O.lib().orc_kernel_basis(n, C.c_void_p(v.ctypes.data), C.c_void_p(Nn.ctypes.data))
            assert np.abs(Nn.T @ Nn - np.eye(n - 1)).max() < 1e-14 and np.abs(Nn.T @ v).max() < 1e-14


def test_pcg_termination_semantics(small_problem):
    """ConjugateGradientsSolver::solve exits (conjugate_gradient.hpp:131-136, 269-301) as restated."""
    from oracle import povar_oracle as O
    p = small_problem
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    lms = orc.init_landmarks_pose(0.01, p.cams)
    st, ok = orc.linearize_pose(0.01, p.cams, lms)
    sigma = 1.0 / (1e-5 + np.sqrt(orc.jp_diag2_pose(st)))
    orc.scale_jp_cols_pose(st, sigma)
    S, b = orc.get_hb_pose(st, 1.0)
    minv = orc.block_jacobi_inverse(S, 12)
    # |b| = 0: immediate success with a zero step
    x, it, status = orc.pcg(S, np.zeros_like(b), minv)
    assert it == 0 and status == 1 and not x.any()
    # max_iterations reached without the zeta test firing: NO_CONVERGENCE
    x, it, status = orc.pcg(S, b, minv, eta=0.0, max_iterations=3)
    assert it == 3 and status == 0
    # zeta test: i * (Q_i - Q_{i-1}) / Q_i < eta at the returned iteration and not before
    x, it, status = orc.pcg(S, b, minv, eta=1e-3)
    assert status == 1 and 1 < it < 100
    Q = lambda v: v @ S @ v - 2 * b @ v  # noqa: E731
    xs = [np.zeros_like(b)] + [-orc.pcg(S, b, minv, eta=0.0, max_iterations=k)[0] for k in range(1, it + 1)]
    zeta = [k * (Q(xs[k]) - Q(xs[k - 1])) / Q(xs[k]) for k in range(1, it + 1)]
    assert zeta[-1] < 1e-3 and all(z >= 1e-3 for z in zeta[:-1])
    # min_iterations overrides the test; a tight forcing sequence reaches the exact solve
    x2, it2, _ = orc.pcg(S, b, minv, eta=1e-3, min_iterations=it + 3)
    assert it2 >= it + 3
    x3, it3, st3 = orc.pcg(S, b, minv, eta=1e-14)
    assert rel(x3, np.linalg.solve(S, -b)) < 1e-7
    # without a preconditioner (z = r) CG still converges, more slowly
    x4, it4, st4 = orc.pcg(S, b, None, eta=1e-14, max_iterations=2000)
    assert rel(x4, x3) < 1e-6 and it4 > it3
    # not positive definite: p'q <= 0 -> NO_CONVERGENCE; the direct solve reports it
    x5, it5, st5 = orc.pcg(-S, b, None, eta=1e-14)
    assert st5 == 0 and it5 == 1
    assert orc.cholesky_solve(-S, b)[1] == 1


def _permute_landmarks(p, perm):
    """The same problem with its landmarks listed in another order (observations stay grouped by landmark,
    cameras ascending inside each)."""
    k = np.diff(p.lm_off)
    lm_off = np.concatenate([[0], np.cumsum(k[perm])]).astype(np.int32)
    idx = np.concatenate([np.arange(p.lm_off[l], p.lm_off[l + 1]) for l in perm])
    return lm_off, p.cam_idx[idx], p.obs[idx]


def test_landmark_order_invariance_hypothesis():
    """permuting the landmark order changes only the order of the per-camera sums, so
    every per-camera quantity agrees to reduction tolerance and per-landmark ones are permuted copies."""
    from hypothesis import given, settings, strategies as hst
    from oracle import povar_oracle as O
    from povar_amd import synth

    @settings(max_examples=6, deadline=None)
    @given(seed=hst.integers(0, 10_000), n_c=hst.integers(4, 12), n_l=hst.integers(30, 120))
    def check(seed, n_c, n_l):
        p = synth.make_problem(n_c, n_l, 4 * n_l, seed=seed)
        perm = np.random.default_rng(seed).permutation(p.n_lms)
        lm_off2, cam2, obs2 = _permute_landmarks(p, perm)
        a = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
        b = O.Oracle(p.n_cams, lm_off2, cam2, obs2)
        lms_a = a.init_landmarks_pose(0.01, p.cams)
        lms_b = b.init_landmarks_pose(0.01, p.cams)
        assert np.array_equal(lms_b, lms_a[perm])
        res = []
        for orc, lms in ((a, lms_a), (b, lms_b)):
            st, diag2, jls, sigma, ok = orc.stage1_pose(0.01, p.cams, lms)
            orc.scale_jp_cols_pose(st, sigma)
            hll, bb, binv = orc.prepare_hb_pose(st, 1e-2)
            inc, it, status, _ = orc.solve_pose(st, hll, binv, bb, 10)
            S, b2 = orc.get_hb_pose(st, 1e-2)
            x_pcg, it_pcg, st_pcg = orc.pcg(S, b2, orc.block_jacobi_inverse(S, 12), eta=1e-3)
            res.append((sigma, bb, inc, x_pcg, it_pcg, orc.error_pose(0.01, p.cams, lms).all_error))
        for u, v in zip(res[0][:4], res[1][:4]):
            assert rel(v, u) < 1e-9
        assert res[0][4] == res[1][4] and abs(res[0][5] - res[1][5]) <= 1e-12 * res[0][5]

    check()
