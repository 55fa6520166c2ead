"""Seeded random sweep of the whole C ABI against the CPU oracle: problem shape (including single-camera-heavy
graphs, two-view landmarks and >64-observation landmarks), robust norm, damping, series length, E0 mode and
linear solver are drawn per case; every stage of one LM inner iteration is compared."""
import os

import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu
ALPHA = 0.01


def _random_problem(rng):
    n_c = int(rng.integers(3, 90))
    n_l = int(rng.integers(20, 400))
    degs = np.minimum(2 + rng.geometric(rng.uniform(0.15, 0.6), size=n_l), n_c)
    if n_c > 70 and rng.random() < 0.5:
        degs[: int(rng.integers(1, 4))] = int(rng.integers(65, n_c))      # long landmarks (lm_long driver)
    pop = 1.0 / (1 + np.arange(n_c)) ** rng.uniform(0.0, 1.5)
    pop /= pop.sum()
    cam_idx = np.concatenate([np.sort(rng.choice(n_c, int(k), replace=False, p=pop)) for k in degs]).astype(np.int32)
    lm_off = np.concatenate([[0], np.cumsum(degs)]).astype(np.int32)
    cams = np.zeros((n_c, 12))
    cams[:, :8] = rng.normal(size=(n_c, 8))
    cams[:, 11] = 1.0
    X = rng.normal(size=(n_l, 3))
    lm_of = np.repeat(np.arange(n_l), degs)
    P = cams[cam_idx].reshape(-1, 3, 4)
    obs = np.einsum("nij,nj->ni", P[:, :2, :3], X[lm_of]) + P[:, :2, 3] + rng.normal(scale=0.1, size=(len(cam_idx), 2))
    return n_c, lm_off, cam_idx, obs, cams


@pytest.mark.parametrize("seed", range(12))
def test_step1_random_case(seed):
    from povar_amd import capi
    from oracle import povar_oracle as O
    rng = np.random.default_rng(1000 + seed)
    n_c, lm_off, cam_idx, obs, cams = _random_problem(rng)
    norm = ["NONE", "HUBER", "CAUCHY"][int(rng.integers(0, 3))]
    huber = float(rng.uniform(0.05, 2.0))
    lam = float(10 ** rng.uniform(-5, 1))
    m = int(rng.integers(0, 26))
    mode = int(rng.integers(0, 4))
    solver = int(rng.integers(0, 2))           # POWER_VARPROJ / POWER_SCHUR_COMPLEMENT
    orc = O.Oracle(n_c, lm_off, cam_idx, obs, robust_norm=norm, huber=huber)
    ctx = capi.Context(n_c, lm_off, cam_idx, obs, robust_norm=norm, huber=huber, e0_mode=mode)
    ctx.set_cameras(cams)
    ctx.init_landmarks_pose(ALPHA)
    lms = orc.init_landmarks_pose(ALPHA, cams)
    assert rel(ctx.get_landmarks(), lms) < 1e-9
    ctx.set_landmarks(lms)
    ri, ro = ctx.error_pose(ALPHA), orc.error_pose(ALPHA, cams, lms)
    assert abs(ri.all_error - ro.all_error) <= 1e-11 * ro.all_error
    assert ctx.linearize_pose(ALPHA)
    st, diag2, jls, sigma, ok = orc.stage1_pose(ALPHA, cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    lam_lm = lam if solver == capi.POWER_SCHUR_COMPLEMENT else 0.0
    hll, b, binv = orc.prepare_hb_pose(st, lam, lam_lm)
    ref, it_o, st_o, _ = orc.solve_pose(st, hll, binv, b, m)
    inc, it, status, rc = ctx.solve_pose(lam, solver, m)
    assert rc == 0 and it == it_o and rel(inc, ref) < 1e-9, (norm, lam, m, mode, solver)
    if solver == capi.POWER_VARPROJ:
        inc_s = ref * sigma
        cams_new = cams + inc_s.reshape(-1, 12)
        ld_o, lms_new = orc.back_substitute_pose(ALPHA, st, cams_new, lms, inc_s * (1.0 / sigma))
    else:
        ld_o, lms_new = orc.back_substitute_poba(st, jls, lam, lms, ref)
        cams_new = cams + (ref * sigma).reshape(-1, 12)
    ld = ctx.apply_pose(solver, ALPHA, ref)
    assert abs(ld - ld_o) <= 1e-8 * abs(ld_o) + 1e-300
    assert rel(ctx.get_cameras(), cams_new) < 1e-13 and rel(ctx.get_landmarks(), lms_new) < 1e-8
    # the explicit-SC solvers on the same linearisation (PCG truncated at a random count, CHOLESKY)
    ctx.set_cameras(cams)
    ctx.set_landmarks(lms)
    ctx.set_jl_col_scaling(False)
    assert ctx.linearize_pose(ALPHA)
    st2, ok = orc.linearize_pose(ALPHA, cams, lms)
    orc.scale_jp_cols_pose(st2, 1.0 / (1e-5 + np.sqrt(orc.jp_diag2_pose(st2))))
    S, b2 = orc.get_hb_pose(st2, lam)
    k = int(rng.integers(1, 9))
    ref_k, it_k, st_k = orc.pcg(S, b2, orc.block_jacobi_inverse(S, 12), eta=0.0, max_iterations=k)
    inc_k, it_g, st_g, rc = ctx.solve_pose_sc(lam, capi.SC_PCG, 0, k, 0.0)
    if st_k == 0 and it_k == k:     # (a breakdown exit at the noise level may fall on either side)
        assert rc == 0 and (it_g, st_g) == (it_k, st_k) and rel(inc_k, ref_k) < 1e-7
    ref_c, bad = orc.cholesky_solve(S, b2)
    inc_c, _, _, rc = ctx.solve_pose_sc(lam, capi.SC_CHOLESKY)
    assert bad == 0 and rc == 0 and rel(inc_c, ref_c) < 1e-7
    ctx.close()


@pytest.mark.parametrize("seed", range(6))
def test_step2_random_case(seed):
    from povar_amd import capi
    from oracle import povar_oracle as O
    rng = np.random.default_rng(2000 + seed)
    n_c, lm_off, cam_idx, obs, _ = _random_problem(rng)
    n_l = lm_off.shape[0] - 1
    cams = rng.normal(size=(n_c, 12))
    cams[:, 8:11] *= 0.1
    cams[:, 11] = 5 + rng.random(n_c)
    cams /= np.linalg.norm(cams, axis=1, keepdims=True)
    lms_h = np.concatenate([rng.normal(size=(n_l, 3)), np.ones((n_l, 1))], 1)
    obs = obs / 50.0
    norm = ["NONE", "HUBER"][int(rng.integers(0, 2))]
    lam = float(10 ** rng.uniform(-5, 0))
    m = int(rng.integers(0, 16))
    mode = [0, 2][int(rng.integers(0, 2))]
    orc = O.Oracle(n_c, lm_off, cam_idx, obs, robust_norm=norm, huber=0.5)
    ctx = capi.Context(n_c, lm_off, cam_idx, obs, robust_norm=norm, huber=0.5, e0_mode=mode)
    ctx.set_cameras(cams)
    ctx.set_landmarks_homogeneous(lms_h)
    assert ctx.linearize_homogeneous()
    st_h, ok = orc.linearize_homogeneous(cams, lms_h)
    diag2 = orc.jp_diag2_homogeneous(st_h)
    jls = orc.scale_jl_cols_homogeneous(st_h)
    sigma = 1.0 / (1e-5 + np.sqrt(diag2))
    orc.scale_jp_cols_joint(st_h, sigma)
    st_n = orc.linearize_nullspace(cams, lms_h, st_h)
    hll, b, binv = orc.prepare_hb_joint(st_h, st_n, lam)
    ref, it_o, st_o, _ = orc.solve_joint(st_n, hll, binv, b, m)
    inc, it, status, rc = ctx.solve_joint(lam, m)
    assert rc == 0 and it == it_o and rel(inc, ref) < 1e-8, (norm, lam, m, mode)
    S, b2 = orc.get_hb_joint(st_h, st_n, lam)
    k = int(rng.integers(1, 7))
    ref_k, it_k, st_k = orc.pcg(S, b2, orc.block_jacobi_inverse(S, 11), dim=11, eta=0.0, max_iterations=k)
    inc_k, it_g, st_g, rc = ctx.solve_joint_sc(lam, 0, k, 0.0)
    if st_k == 0 and it_k == k:
        assert rc == 0 and (it_g, st_g) == (it_k, st_k) and rel(inc_k, ref_k) < 1e-6
    ld = ctx.apply_joint(ref)
    ld_o, lms_new = orc.back_substitute_joint(st_h, jls, lam, cams, lms_h, ref)
    assert abs(ld - ld_o) <= 1e-8 * abs(ld_o) + 1e-300
    assert rel(ctx.get_landmarks_homogeneous(), lms_new) < 1e-8
    ctx.close()


@pytest.mark.parametrize("hot_acc,long_separate", [(4, False), (4, True), (40, False), (1000, False)])
def test_small_hot_set_with_long_landmarks(hot_acc, long_separate, monkeypatch):
    """The LDS-accumulation modes with a hot set smaller than the camera count (POVAR_HOT_ACC) on a problem with
    long landmarks: cold cameras, cold observations on long landmarks and LDS-accumulated observations on long
    landmarks all occur, with the long landmarks walked inside e0_lm_cached<true> or by the lm_long kernel
    (POVAR_LONG_SEPARATE).  Same answers as the oracle in every combination, steps 1 and 2, PCG included."""
    from povar_amd import capi
    from oracle import povar_oracle as O
    monkeypatch.setenv("POVAR_HOT_ACC", str(hot_acc))
    if long_separate:
        monkeypatch.setenv("POVAR_LONG_SEPARATE", "1")
    rng = np.random.default_rng(77)
    n_c = 120
    degs = np.array([110, 97, 65, 64, 70] + [2] * 60 + list(rng.integers(3, 12, size=220)))
    degs = np.minimum(degs, n_c)
    pop = 1.0 / (1 + np.arange(n_c)) ** 0.8
    pop /= pop.sum()
    cam_idx = np.concatenate([np.sort(rng.choice(n_c, int(k), replace=False, p=pop)) for k in degs]).astype(np.int32)
    lm_off = np.concatenate([[0], np.cumsum(degs)]).astype(np.int32)
    n_l = len(degs)
    cams = np.zeros((n_c, 12))
    cams[:, :8] = rng.normal(size=(n_c, 8))
    cams[:, 11] = 1.0
    X = rng.normal(size=(n_l, 3))
    lm_of = np.repeat(np.arange(n_l), degs)
    P = cams[cam_idx].reshape(-1, 3, 4)
    obs = np.einsum("nij,nj->ni", P[:, :2, :3], X[lm_of]) + P[:, :2, 3] + rng.normal(scale=0.05, size=(len(cam_idx), 2))
    orc = O.Oracle(n_c, lm_off, cam_idx, obs)
    lms = orc.init_landmarks_pose(ALPHA, cams)
    st, diag2, jls, sigma, ok = orc.stage1_pose(ALPHA, cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    lam, m = 1e-3, 12
    hll, b, binv = orc.prepare_hb_pose(st, lam)
    ref, _, _, terms = orc.solve_pose(st, hll, binv, b, m, want_terms=True)
    for mode in (capi.E0_IMPLICIT_LDSACC, capi.E0_TILES_LDSACC):
        ctx = capi.Context(n_c, lm_off, cam_idx, obs, e0_mode=mode)
        ctx.set_cameras(cams)
        ctx.set_landmarks(lms)
        assert ctx.linearize_pose(ALPHA)
        ctx.prepare_pose(lam)
        ctx.power_series_begin()
        for i in range(1, m + 1):
            ctx.power_series_step()
            assert rel(ctx.get_term(), terms[i]) < 1e-10, (mode, i)
        inc, it, status, rc = ctx.solve_pose(lam, capi.POWER_VARPROJ, m)
        assert rc == 0 and rel(inc, ref) < 1e-10
        x = rng.normal(size=12 * n_c)
        assert rel(ctx.right_mul_e0_pose(x), orc.right_mul_e0_pose(st, hll, x)) < 1e-11
        pcg_a, it_a, st_a, _ = ctx.solve_pose_sc(lam, capi.SC_PCG, 0, 6, 0.0)
        ctx.set_e0_mode(capi.E0_IMPLICIT)
        pcg_b, it_b, st_b, _ = ctx.solve_pose_sc(lam, capi.SC_PCG, 0, 6, 0.0)
        assert (it_a, st_a) == (it_b, st_b) and rel(pcg_a, pcg_b) < 1e-9
        ctx.close()
    # step 2 with the same structure
    rng2 = np.random.default_rng(5)
    cams2 = rng2.normal(size=(n_c, 12))
    cams2[:, 8:11] *= 0.1
    cams2[:, 11] = 5 + rng2.random(n_c)
    cams2 /= np.linalg.norm(cams2, axis=1, keepdims=True)
    lms_h = np.concatenate([rng2.normal(size=(n_l, 3)), np.ones((n_l, 1))], 1)
    obs2 = obs / 50.0
    orc2 = O.Oracle(n_c, lm_off, cam_idx, obs2)
    st_h, ok = orc2.linearize_homogeneous(cams2, lms_h)
    diag2 = orc2.jp_diag2_homogeneous(st_h)
    orc2.scale_jl_cols_homogeneous(st_h)
    orc2.scale_jp_cols_joint(st_h, 1.0 / (1e-5 + np.sqrt(diag2)))
    st_n = orc2.linearize_nullspace(cams2, lms_h, st_h)
    hll2, b2, binv2 = orc2.prepare_hb_joint(st_h, st_n, lam)
    ref2, _, _, _ = orc2.solve_joint(st_n, hll2, binv2, b2, 8)
    ctx = capi.Context(n_c, lm_off, cam_idx, obs2, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.set_cameras(cams2)
    ctx.set_landmarks_homogeneous(lms_h)
    assert ctx.linearize_homogeneous()
    inc2, it2, st2, rc = ctx.solve_joint(lam, 8)
    assert rc == 0 and rel(inc2, ref2) < 1e-8
    ctx.close()


def test_init_landmarks_near_degenerate():
    """K1 on two-view landmarks whose four camera rows are nearly coplanar (kappa(G) up to ~1e7): the reference
    solves G x = z with bdcSvd (HLP:94), i.e. with an error ~ kappa u; the QR kernel (init_landmarks_qr) keeps that,
    the round-1 normal-equation kernels (POVAR_K1_NORMAL_EQ=1: kappa^2 u, one refinement step) do not."""
    import os
    from povar_amd import capi
    from oracle import povar_oracle as O
    rng = np.random.default_rng(17)
    n_pairs, per_pair = 60, 20
    n_c = 2 * n_pairs
    cams = np.zeros((n_c, 12))
    eps = 10.0 ** rng.uniform(-7.2, -1.0, size=n_pairs)
    for i in range(n_pairs):
        A = rng.normal(size=(2, 4))
        M = rng.normal(size=(2, 2))
        B = M @ A + eps[i] * rng.normal(size=(2, 4))
        B[:, 3] = rng.normal(size=2)              # the translation column is free
        cams[2 * i, :8] = A.ravel()
        cams[2 * i + 1, :8] = B.ravel()
    cams[:, 11] = 1.0
    n_l = n_pairs * per_pair
    lm_off = (2 * np.arange(n_l + 1)).astype(np.int32)
    cam_idx = np.repeat(np.arange(n_pairs), per_pair)[:, None] * 2 + np.array([0, 1])[None, :]
    cam_idx = cam_idx.ravel().astype(np.int32)
    obs = rng.normal(scale=30.0, size=(2 * n_l, 2))
    alpha = 0.01
    ref = O.Oracle(n_c, lm_off, cam_idx, obs).init_landmarks_pose(alpha, cams)
    # condition number of every landmark's G (8 x 3)
    sa, sb = np.sqrt(alpha), np.sqrt(1 - alpha)
    kappa = np.zeros(n_l)
    for l in range(n_l):
        rows = []
        for o in (2 * l, 2 * l + 1):
            P = cams[cam_idx[o]].reshape(3, 4)
            u, v = obs[o]
            rows += [sb * (P[0, :3] - u * P[2, :3]), sb * (P[1, :3] - v * P[2, :3]), sa * P[0, :3], sa * P[1, :3]]
        s = np.linalg.svd(np.array(rows), compute_uv=False)
        kappa[l] = s[0] / s[-1]
    assert kappa.max() > 1e6 and kappa.min() < 1e3

    def run_x(env):
        old = os.environ.get("POVAR_K1_NORMAL_EQ")
        if env:
            os.environ["POVAR_K1_NORMAL_EQ"] = "1"
        try:
            ctx = capi.Context(n_c, lm_off, cam_idx, obs)
            ctx.set_cameras(cams)
            ctx.init_landmarks_pose(alpha)
            x = ctx.get_landmarks()
            ctx.close()
        finally:
            if env:
                if old is None:
                    del os.environ["POVAR_K1_NORMAL_EQ"]
                else:
                    os.environ["POVAR_K1_NORMAL_EQ"] = old
        return x

    # Ground truth: the exact least-squares solution of the fp64 system (rational arithmetic on the normal
    # equations).  With noisy observations the residual is not small, so the PROBLEM is conditioned like
    # kappa + kappa^2 |r| / (|G||x|): two backward-stable solvers may differ by that much, which is why the bar is the
    # oracle's own distance to the truth, per decade of kappa, and not a fixed number.
    from fractions import Fraction as Fr
    exact = np.zeros((n_l, 3))
    for l in range(n_l):
        rows, zz = [], []
        for o in (2 * l, 2 * l + 1):
            P = cams[cam_idx[o]].reshape(3, 4)
            u, v = obs[o]
            rows += [sb * (P[0, :3] - u * P[2, :3]), sb * (P[1, :3] - v * P[2, :3]), sa * P[0, :3], sa * P[1, :3]]
            zz += [sb * (P[2, 3] * u - P[0, 3]), sb * (P[2, 3] * v - P[1, 3]), sa * (u - P[0, 3]), sa * (v - P[1, 3])]
        Gq = [[Fr(float(x)) for x in r] for r in rows]
        zq = [Fr(float(x)) for x in zz]
        H = [[sum(Gq[r][a] * Gq[r][b] for r in range(8)) for b in range(3)] for a in range(3)]
        g = [sum(Gq[r][a] * zq[r] for r in range(8)) for a in range(3)]
        det = (H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) - H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0])
               + H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]))

        def col(k):
            Mx = [[g[i] if j == k else H[i][j] for j in range(3)] for i in range(3)]
            return (Mx[0][0] * (Mx[1][1] * Mx[2][2] - Mx[1][2] * Mx[2][1]) - Mx[0][1] * (Mx[1][0] * Mx[2][2] - Mx[1][2] * Mx[2][0])
                    + Mx[0][2] * (Mx[1][0] * Mx[2][1] - Mx[1][1] * Mx[2][0]))
        exact[l] = [float(col(k) / det) for k in range(3)]
    nrm = np.linalg.norm(exact, axis=1)
    err_orc = np.linalg.norm(ref - exact, axis=1) / nrm

    def dist(x):
        return np.linalg.norm(x - exact, axis=1) / nrm

    err = dist(run_x(False))
    err_ne = dist(run_x(True))
    for lo in (1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7):
        sel = (kappa >= lo) & (kappa < 10 * lo)
        if sel.sum() < 5:
            continue
        # as accurate as the reference's class of solver in every decade of kappa ...
        assert err[sel].max() <= 10 * err_orc[sel].max() + 1e-14, (lo, err[sel].max(), err_orc[sel].max())
    # ... while the normal equations lose the ill-conditioned landmarks by orders of magnitude (the test bites)
    bad = kappa > 1e6
    assert np.median(err_ne[bad]) > 100 * np.median(err[bad])


def test_device_timings():
    """povar_timings: hipEvent stage timings per entry point (the IterationSummary timing fields)."""
    from povar_amd import capi, synth
    p = synth.make_problem(49, 2000, 8200, seed=7)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.set_cameras(p.cams)
    ctx.timings_enable(True)
    ctx.init_landmarks_pose(0.01)
    ctx.error_pose(0.01)
    assert ctx.linearize_pose(0.01)
    for _ in range(3):
        inc, it, st, rc = ctx.solve_pose(1e-4, capi.POWER_VARPROJ, 20)
    ctx.apply_pose(capi.POWER_VARPROJ, 0.01, inc)
    t = ctx.timings()
    assert (t.linearize_calls, t.prepare_calls, t.solve_calls, t.apply_calls, t.other_calls) == (1, 3, 3, 1, 2)
    for ms in (t.linearize_ms, t.prepare_ms, t.solve_ms, t.apply_ms, t.other_ms):
        assert 0 < ms < 1000
    assert t.solve_ms / 3 > 20 * 0.004            # 20 terms of at least two kernels each
    t2 = ctx.timings()                             # accumulates; nothing new happened
    assert t2.solve_ms == t.solve_ms and t2.solve_calls == 3
    ctx.timings_enable(False)
    ctx.solve_pose(1e-4, capi.POWER_VARPROJ, 20)
    assert ctx.timings().solve_calls == 0
    ctx.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("POVAR_FUZZ_SEQ_SEEDS", "6"))))
def test_api_sequence_differential(seed, monkeypatch):
    """Random sequences of C-ABI calls on two contexts over the same problem: A in the shipping mode with the
    lane-per-landmark kernels forced (lazy legacy arrays, mode switches in the middle of an iteration, exports),
    B in the bit-reproducible POVAR_E0_IMPLICIT mode (every legacy array written eagerly).  Whatever the order of the
    calls, both must return the same numbers: a stale lazily-built array would show up here."""
    from povar_amd import capi
    monkeypatch.setenv("POVAR_E0_V1", "0")
    rng = np.random.default_rng(7000 + seed)
    n_c, lm_off, cam_idx, obs, cams = _random_problem(rng)
    norm = ["NONE", "HUBER", "CAUCHY"][int(rng.integers(0, 3))]
    kw = dict(robust_norm=norm, huber=float(rng.uniform(0.1, 2.0)))
    A = capi.Context(n_c, lm_off, cam_idx, obs, e0_mode=capi.E0_IMPLICIT_LDSACC, **kw)
    B = capi.Context(n_c, lm_off, cam_idx, obs, e0_mode=capi.E0_IMPLICIT, **kw)
    assert A.layout_info().lane_per_landmark == 1
    for c in (A, B):
        c.set_cameras(cams)
        c.init_landmarks_pose(ALPHA)
    B.set_landmarks(A.get_landmarks())
    both = lambda f: (f(A), f(B))  # noqa: E731
    linearized = prepared = False
    solver = capi.POWER_VARPROJ
    inc = None
    modes = [capi.E0_IMPLICIT_LDSACC, capi.E0_IMPLICIT, capi.E0_TILES, capi.E0_TILES_LDSACC]
    log = []
    for step in range(int(os.environ.get("POVAR_FUZZ_SEQ_LEN", "40"))):
        ops = ["error", "linearize", "mode"]
        if linearized:
            ops += ["prepare", "export_lin"]
        if prepared:
            ops += ["series", "e0", "export_prep", "series"]
        if prepared and inc is not None:
            ops += ["apply"]
        op = ops[int(rng.integers(0, len(ops)))]
        log.append(op)
        if op == "error":
            ra, rb = both(lambda c: c.error_pose(ALPHA))
            assert ra.all_num_obs == rb.all_num_obs and abs(ra.all_error - rb.all_error) <= 1e-12 * abs(rb.all_error), log
        elif op == "linearize":
            oa, ob = both(lambda c: c.linearize_pose(ALPHA))
            assert oa and ob
            linearized, prepared, inc = True, False, None
        elif op == "mode":
            A.set_e0_mode(modes[int(rng.integers(0, 4))] if rng.random() < 0.5 else capi.E0_IMPLICIT_LDSACC)
        elif op == "prepare":
            lam = float(10 ** rng.uniform(-4, 0))
            solver = capi.POWER_VARPROJ if rng.random() < 0.7 else capi.POWER_SCHUR_COMPLEMENT
            both(lambda c: c.prepare_pose(lam, solver))
            prepared, inc = True, None
        elif op == "export_lin":
            for which in (capi.BUF_DIAG2, capi.BUF_POSE_SCALING, capi.BUF_JL_COL_SCALE):
                a, b = both(lambda c: c.get_buffer(which))
                assert rel(a, b) < 1e-12, (log, which)
        elif op == "export_prep":
            for which, tol in ((capi.BUF_HLL_INV, 1e-8), (capi.BUF_B, 1e-8), (capi.BUF_B_INV, 1e-8), (capi.BUF_STORAGE, 1e-12)):
                a, b = both(lambda c: c.get_buffer(which))
                assert rel(a, b) < tol, (log, which)
        elif op == "series":
            m = int(rng.integers(0, 12))
            both(lambda c: c.power_series_pose(m))
            ia, ib = both(lambda c: c.get_increment())
            assert rel(ia, ib) < 1e-8, log
            inc = ib
        elif op == "e0":
            x = rng.normal(size=12 * n_c)
            ya, yb = both(lambda c: c.right_mul_e0_pose(x))
            assert rel(ya, yb) < 1e-11, log
        elif op == "apply":
            both(lambda c: c.backup_pose())
            la, lb = both(lambda c: c.apply_pose(solver, ALPHA, inc))
            assert abs(la - lb) <= 1e-8 * max(abs(lb), 1e-300), log
            ca, cb = both(lambda c: c.get_cameras())
            xa, xb = both(lambda c: c.get_landmarks())
            assert rel(ca, cb) < 1e-12 and rel(xa, xb) < 1e-8, log
            if rng.random() < 0.5:  # rejected step: back to the linearisation point, the prepared system stays valid
                both(lambda c: c.restore_pose())
            else:
                B.set_landmarks(xa)  # accepted: new state, identical in both
                B.set_cameras(ca)
                A.set_landmarks(xa)
                A.set_cameras(ca)
                linearized = prepared = False
            inc = None
    A.close()
    B.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("POVAR_FUZZ_SEQ_SEEDS", "4"))))
def test_api_sequence_differential_step2(seed, monkeypatch):
    """The same differential sequence test for the projective refinement (linearize_homogeneous / prepare_joint /
    series / apply_joint / normalize_joint): lane-per-landmark shipping mode against POVAR_E0_IMPLICIT."""
    from povar_amd import capi
    monkeypatch.setenv("POVAR_E0_V1", "0")
    rng = np.random.default_rng(9000 + seed)
    n_c, lm_off, cam_idx, obs, _ = _random_problem(rng)
    n_l = len(lm_off) - 1
    cams = rng.normal(size=(n_c, 12))
    cams[:, 8:11] *= 0.1
    cams[:, 11] = 5 + rng.random(n_c)
    cams /= np.linalg.norm(cams, axis=1, keepdims=True)
    lms_h = np.concatenate([rng.normal(size=(n_l, 3)), np.ones((n_l, 1))], 1)
    obs = obs / 50.0
    norm = ["NONE", "HUBER"][int(rng.integers(0, 2))]
    kw = dict(robust_norm=norm, huber=float(rng.uniform(0.01, 0.5)))
    A = capi.Context(n_c, lm_off, cam_idx, obs, e0_mode=capi.E0_IMPLICIT_LDSACC, **kw)
    B = capi.Context(n_c, lm_off, cam_idx, obs, e0_mode=capi.E0_IMPLICIT, **kw)
    for c in (A, B):
        c.set_cameras(cams)
        c.set_landmarks_homogeneous(lms_h)
    both = lambda f: (f(A), f(B))  # noqa: E731
    linearized = prepared = False
    inc = None
    log = []
    for step in range(int(os.environ.get("POVAR_FUZZ_SEQ_LEN", "40"))):
        ops = ["error", "linearize", "mode"]
        if linearized:
            ops += ["prepare"]
        if prepared:
            ops += ["series", "export_prep", "series"]
        if prepared and inc is not None:
            ops += ["apply"]
        op = ops[int(rng.integers(0, len(ops)))]
        log.append(op)
        if op == "error":
            ra, rb = both(lambda c: c.error_homogeneous())
            assert ra.valid_num_obs == rb.valid_num_obs and abs(ra.all_error - rb.all_error) <= 1e-12 * abs(rb.all_error), log
        elif op == "linearize":
            oa, ob = both(lambda c: c.linearize_homogeneous())
            assert oa and ob
            linearized, prepared, inc = True, False, None
        elif op == "mode":
            A.set_e0_mode(capi.E0_IMPLICIT if rng.random() < 0.4 else capi.E0_IMPLICIT_LDSACC)
        elif op == "prepare":
            lam = float(10 ** rng.uniform(-4, 0))
            both(lambda c: c.prepare_joint(lam))
            prepared, inc = True, None
        elif op == "export_prep":
            # (tolerances: the two modes sum in different orders, and a random projective state after several applies is badly
            # conditioned -- observed up to 3e-8; a stale lazily-built array would be O(1e-2) or more off)
            for which, tol in ((capi.BUF_HLL_INV, 1e-6), (capi.BUF_B_JOINT, 1e-6), (capi.BUF_B_INV_JOINT, 1e-6),
                               (capi.BUF_JL_COL_SCALE_H, 1e-12), (capi.BUF_DIAG2, 1e-12)):
                a, b = both(lambda c: c.get_buffer(which))
                assert rel(a, b) < tol, (log, which)
        elif op == "series":
            m = int(rng.integers(0, 10))
            both(lambda c: c.power_series_pose(m))
            ia, ib = both(lambda c: c.get_increment(11))
            assert rel(ia, ib) < 1e-6, log
            inc = ib
        elif op == "apply":
            both(lambda c: c.backup_joint())
            la, lb = both(lambda c: c.apply_joint(inc))
            assert abs(la - lb) <= 1e-6 * max(abs(lb), 1e-300), log
            ca, cb = both(lambda c: c.get_cameras())
            xa, xb = both(lambda c: c.get_landmarks_homogeneous())
            assert rel(ca, cb) < 1e-10 and rel(xa, xb) < 1e-6, log
            if rng.random() < 0.5:
                both(lambda c: c.restore_joint())
            else:
                both(lambda c: c.normalize_joint())
                ca, xa = A.get_cameras(), A.get_landmarks_homogeneous()
                for c in (A, B):
                    c.set_cameras(ca)
                    c.set_landmarks_homogeneous(xa)
                linearized = prepared = False
            inc = None
    A.close()
    B.close()
