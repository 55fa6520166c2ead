"""Independent dense NumPy restatement of the PoVar power-series path -- TEST INFRASTRUCTURE.

Purpose: a second, structurally different implementation (global dense Jacobians +
numpy.linalg) that pins the C oracle (oracle/povar_oracle.c) in the absence of any
reference-side golden vectors ("parity unpinned", see oracle/povar_oracle.h).  Used only in
this container by tests/golden/make_golden.py to generate the committed fixtures and by the
``-m "not gpu"`` tests; small problems only (everything is dense).

Follows SURVEY.md Appendix A (reference: /root/reference/src/rootba_povar/):
  A.1 residual/Jacobians  bal/bal_bundle_adjustment_helper.cpp:244-313
  A.2 robust weight       bal/bal_bundle_adjustment_helper.cpp:52-74
  A.3 linearise           solver/linearizor_power_varproj.cpp:45-76
  A.4 solve               solver/linearizor_power_varproj.cpp:178-243,
                          sc/linearization_power_varproj.hpp:124-237, 364-406
  A.5 apply               solver/linearizor_power_varproj.cpp:246-273, sc/landmark_block.hpp:625-707
  A.7 step 2              bal/bal_bundle_adjustment_helper.cpp:316-380, sc/landmark_block.hpp:180-269,
                          474-507, 574-623, sc/linearization_power_varproj.hpp:74-122, 240-287, 408-453
"""
from __future__ import annotations

import numpy as np
import scipy.linalg


def _weights(r2, norm, huber):
    if norm == "HUBER":
        w = np.where(r2 < huber * huber, 1.0, huber / np.sqrt(np.maximum(r2, 1e-300)))
        e = 0.5 * (2 - w) * w * r2
    elif norm == "CAUCHY":
        w = np.ones_like(r2)
        e = np.log1p(r2)
    else:
        w = np.ones_like(r2)
        e = 0.5 * r2
    return e, w


def _A(alpha, u, v):
    a, ab = np.sqrt(alpha), np.sqrt(1.0 - alpha)
    return np.array([[ab, 0, -ab * u], [0, ab, -ab * v], [a, 0, 0], [0, a, 0]])


def residuals_pose(alpha, lm_off, cam_idx, obs, cams, lms):
    """r_i = A(u,v) P h - (0, 0, a u, a v)   (A.1)"""
    n_o = cam_idx.shape[0]
    lm_of = np.repeat(np.arange(lm_off.shape[0] - 1), np.diff(lm_off))
    a = np.sqrt(alpha)
    r = np.zeros((n_o, 4))
    for i in range(n_o):
        P = cams[cam_idx[i]].reshape(3, 4)
        h = np.append(lms[lm_of[i]], 1.0)
        r[i] = _A(alpha, *obs[i]) @ (P @ h) - np.array([0, 0, a * obs[i, 0], a * obs[i, 1]])
    return r


def dense_pose(alpha, n_cams, lm_off, cam_idx, obs, cams, lms, norm="NONE", huber=1.0):
    """Dense weighted Jacobians of step 1: Jp (4n_o x 12n_c), Jl (4n_o x 3n_l), r (4n_o)."""
    n_o = cam_idx.shape[0]
    n_l = lm_off.shape[0] - 1
    lm_of = np.repeat(np.arange(n_l), np.diff(lm_off))
    r = residuals_pose(alpha, lm_off, cam_idx, obs, cams, lms)
    e, w = _weights((r * r).sum(1), norm, huber)
    Jp = np.zeros((4 * n_o, 12 * n_cams))
    Jl = np.zeros((4 * n_o, 3 * n_l))
    for i in range(n_o):
        c, l = cam_idx[i], lm_of[i]
        A = _A(alpha, *obs[i])
        h = np.append(lms[l], 1.0)
        sw = np.sqrt(w[i])
        Jp[4 * i : 4 * i + 4, 12 * c : 12 * c + 12] = sw * np.kron(A, h[None, :])
        Jl[4 * i : 4 * i + 4, 3 * l : 3 * l + 3] = sw * (A @ cams[c].reshape(3, 4)[:, :3])
        r[i] *= sw
    return Jp, Jl, r.reshape(-1), e, w


def init_landmarks_pose(alpha, lm_off, cam_idx, obs, cams):
    """x_l = argmin sum_i |A P [x;1] - c|^2   (K1; HLP:76-99, 221-241)"""
    n_l = lm_off.shape[0] - 1
    a = np.sqrt(alpha)
    out = np.zeros((n_l, 3))
    for l in range(n_l):
        G, z = [], []
        for i in range(lm_off[l], lm_off[l + 1]):
            M = _A(alpha, *obs[i]) @ cams[cam_idx[i]].reshape(3, 4)
            G.append(M[:, :3])
            z.append(np.array([0, 0, a * obs[i, 0], a * obs[i, 1]]) - M[:, 3])
        out[l] = np.linalg.lstsq(np.vstack(G), np.concatenate(z), rcond=None)[0]
    return out


def step1(alpha, n_cams, lm_off, cam_idx, obs, cams, lms, lam, m, eps=1e-5, norm="NONE",
          huber=1.0, lam_lm=0.0):
    """One linearize + solve + apply of step 1, dense.  Returns a dict of every intermediate."""
    n_l = lm_off.shape[0] - 1
    Jp, Jl, r, e, w = dense_pose(alpha, n_cams, lm_off, cam_idx, obs, cams, lms, norm, huber)
    diag2 = (Jp * Jp).sum(0)
    sigma = 1.0 / (eps + np.sqrt(diag2))
    jl_scale = 1.0 / (eps + np.sqrt((Jl * Jl).sum(0)))
    Jps, Jls = Jp * sigma, Jl * jl_scale
    Hll = Jls.T @ Jls + lam_lm * np.eye(3 * n_l)
    Hll_inv = np.zeros_like(Hll)
    for l in range(n_l):
        s = slice(3 * l, 3 * l + 3)
        Hll_inv[s, s] = np.linalg.inv(Hll[s, s])
    Hpl = Jps.T @ Jls
    E0 = Hpl @ Hll_inv @ Hpl.T
    Hpp = Jps.T @ Jps
    B = Hpp + lam * np.eye(12 * n_cams)
    b = Jps.T @ (r - Jls @ (Hll_inv @ (Jls.T @ r)))
    Binv = np.zeros_like(B)
    for c in range(n_cams):
        s = slice(12 * c, 12 * c + 12)
        Binv[s, s] = np.linalg.inv(B[s, s])
    terms = [Binv @ (-b)]
    for _ in range(m):
        terms.append(Binv @ (E0 @ terms[-1]))
    terms = np.array(terms)
    inc = terms.sum(0)
    exact = np.linalg.solve(B - E0, -b)
    rho = np.abs(np.linalg.eigvals(Binv @ E0)).max()
    return dict(Jp=Jp, Jl=Jl, r=r, diag2=diag2, sigma=sigma, jl_scale=jl_scale.reshape(n_l, 3),
                Jps=Jps, Jls=Jls, hll_inv=np.array([Hll_inv[3 * l : 3 * l + 3, 3 * l : 3 * l + 3]
                                                    for l in range(n_l)]).reshape(n_l, 9),
                b=b, b_inv=np.array([Binv[12 * c : 12 * c + 12, 12 * c : 12 * c + 12]
                                     for c in range(n_cams)]).reshape(n_cams, 144),
                terms=terms, inc=inc, exact=exact, rho=rho, E0=E0, B=B,
                cost=e.sum(), w=w)


def apply_varproj(alpha, n_cams, lm_off, cam_idx, obs, cams, lms, s1, inc):
    """A.5 / A.6: POWER_VARPROJ apply (LZR:250-258, LMB:670-707) with its l_diff quirk."""
    n_l = lm_off.shape[0] - 1
    sigma = s1["sigma"]
    inc_s = inc * sigma
    cams_new = cams + inc_s.reshape(n_cams, 12)
    inc_rt = inc_s * (1.0 / sigma)
    Jp0, Jl0, r0, _, _ = dense_pose(alpha, n_cams, lm_off, cam_idx, obs, cams_new, lms)
    lms_new = lms.copy()
    l_diff = 0.0
    for l in range(n_l):
        rows = slice(4 * lm_off[l], 4 * lm_off[l + 1])
        cols = slice(3 * l, 3 * l + 3)
        J = Jl0[rows, cols]
        delta = -np.linalg.solve(J.T @ J, J.T @ r0[rows])
        Jinc = Jp0[rows] @ inc_rt + s1["Jls"][rows, cols] @ delta
        l_diff -= Jinc @ (0.5 * Jinc + s1["r"][rows])
        lms_new[l] += delta
    return cams_new, lms_new, l_diff


def apply_poba(n_cams, lm_off, cams, lms, s1, inc, lam_lm):
    """POWER_SCHUR_COMPLEMENT apply (LZR:260-270, LMB:625-656)."""
    n_l = lm_off.shape[0] - 1
    lms_new = lms.copy()
    l_diff = 0.0
    for l in range(n_l):
        rows = slice(4 * lm_off[l], 4 * lm_off[l + 1])
        cols = slice(3 * l, 3 * l + 3)
        Jl, Jp, r = s1["Jls"][rows, cols], s1["Jps"][rows], s1["r"][rows]
        jpi = Jp @ inc
        delta = -np.linalg.solve(Jl.T @ Jl + lam_lm * np.eye(3), Jl.T @ (r + jpi))
        Jinc = jpi + Jl @ delta
        l_diff -= Jinc @ (0.5 * Jinc + r)
        lms_new[l] += delta * s1["jl_scale"][l]
    cams_new = cams + (inc * s1["sigma"]).reshape(n_cams, 12)
    return cams_new, lms_new, l_diff


# --------------------------------------------------------------------------- step 2

def dense_homogeneous(n_cams, lm_off, cam_idx, obs, cams, lms_h, norm="NONE", huber=1.0):
    """Dense weighted Jacobians of step 2 in ambient coordinates: Jp (2n_o x 12n_c),
    Jl (2n_o x 4n_l), r (2n_o) (A.7)."""
    n_o = cam_idx.shape[0]
    n_l = lm_off.shape[0] - 1
    lm_of = np.repeat(np.arange(n_l), np.diff(lm_off))
    Jp = np.zeros((2 * n_o, 12 * n_cams))
    Jl = np.zeros((2 * n_o, 4 * n_l))
    r = np.zeros((n_o, 2))
    valid = np.zeros(n_o, dtype=bool)
    for i in range(n_o):
        c, l = cam_idx[i], lm_of[i]
        P = cams[c].reshape(3, 4)
        X = lms_h[l]
        x, y, z = P @ X
        r[i] = np.array([x / z, y / z]) - obs[i]
        valid[i] = abs(z) >= 1e-5
        D = np.array([[1 / z, 0, -x / z**2], [0, 1 / z, -y / z**2]])
        Jp[2 * i : 2 * i + 2, 12 * c : 12 * c + 12] = np.kron(D, X[None, :])
        Jl[2 * i : 2 * i + 2, 4 * l : 4 * l + 4] = D @ P
    e, w = _weights((r * r).sum(1), norm, huber)
    sw = np.sqrt(w)
    Jp *= np.repeat(sw, 2)[:, None]
    Jl *= np.repeat(sw, 2)[:, None]
    r = (r * sw[:, None]).reshape(-1)
    return Jp, Jl, r, e, w, valid


def step2(n_cams, lm_off, cam_idx, obs, cams, lms_h, lam, m, eps=1e-5, norm="NONE", huber=1.0):
    """One linearize + solve_joint + apply_joint of step 2.  Tangent bases come from
    scipy.linalg.null_space (NOT the oracle's Householder basis): every quantity returned is
    ambient or basis-invariant."""
    n_l = lm_off.shape[0] - 1
    Jp, Jl, r, e, w, valid = dense_homogeneous(n_cams, lm_off, cam_idx, obs, cams, lms_h, norm, huber)
    diag2 = (Jp * Jp).sum(0)
    sigma = 1.0 / (eps + np.sqrt(diag2))
    jl_scale = 1.0 / (eps + np.sqrt((Jl * Jl).sum(0)))
    Jps, Jls = Jp * sigma, Jl * jl_scale
    Nc = np.zeros((12 * n_cams, 11 * n_cams))
    for c in range(n_cams):
        Nc[12 * c : 12 * c + 12, 11 * c : 11 * c + 11] = scipy.linalg.null_space(cams[c][None, :])
    Nl = np.zeros((4 * n_l, 3 * n_l))
    for l in range(n_l):
        Nl[4 * l : 4 * l + 4, 3 * l : 3 * l + 3] = scipy.linalg.null_space(lms_h[l][None, :])
    Jpt, Jlt = Jps @ Nc, Jls @ Nl
    Hll = Jlt.T @ Jlt + lam * np.eye(3 * n_l)
    Hll_inv = np.zeros_like(Hll)
    for l in range(n_l):
        s = slice(3 * l, 3 * l + 3)
        Hll_inv[s, s] = np.linalg.inv(Hll[s, s])
    Hpl = Jpt.T @ Jlt
    E0 = Hpl @ Hll_inv @ Hpl.T
    B = Jpt.T @ Jpt + lam * np.eye(11 * n_cams)
    b = Jpt.T @ (r - Jlt @ (Hll_inv @ (Jlt.T @ r)))
    Binv = np.zeros_like(B)
    for c in range(n_cams):
        s = slice(11 * c, 11 * c + 11)
        Binv[s, s] = np.linalg.inv(B[s, s])
    terms = [Binv @ (-b)]
    for _ in range(m):
        terms.append(Binv @ (E0 @ terms[-1]))
    terms = np.array(terms)
    inc = terms.sum(0)
    # apply_joint (LZR:277-308; LMB:574-623)
    pinc = Nc @ inc
    lms_new = lms_h.copy()
    l_diff = 0.0
    for l in range(n_l):
        rows = slice(2 * lm_off[l], 2 * lm_off[l + 1])
        N = Nl[4 * l : 4 * l + 4, 3 * l : 3 * l + 3]
        Jl4 = Jls[rows, 4 * l : 4 * l + 4]
        jlp = Jl4 @ N
        jpi = Jps[rows] @ pinc
        delta = -np.linalg.solve(jlp.T @ jlp + lam * np.eye(3), jlp.T @ (r[rows] + jpi))
        dp = N @ delta
        Jinc = jpi + Jl4 @ dp
        l_diff -= Jinc @ (0.5 * Jinc + r[rows])
        lms_new[l] += dp * jl_scale[4 * l : 4 * l + 4]
    cams_new = cams + (pinc * sigma).reshape(n_cams, 12)
    cams_norm = cams_new / np.linalg.norm(cams_new, axis=1, keepdims=True)
    lms_norm = lms_new / lms_new[:, 3:4]
    return dict(Jp=Jp, Jl=Jl, r=r, diag2=diag2, sigma=sigma, jl_scale=jl_scale.reshape(n_l, 4),
                term_norms=np.linalg.norm(terms, axis=1), ambient_terms=terms @ Nc.T,
                ambient_inc=pinc, l_diff=l_diff, cams_new=cams_new, lms_new=lms_new,
                cams_norm=cams_norm, lms_norm=lms_norm, cost=e.sum(), valid=valid)


def step2_system(n_cams, lm_off, cam_idx, obs, cams, lms_h, lam, eps=1e-5, norm="NONE", huber=1.0):
    """The reduced camera system of step 2 in scipy null_space tangent coordinates: B, E0, b and the
    block-diagonal basis Nc (12 n_cams x 11 n_cams) mapping tangent to ambient vectors."""
    n_l = lm_off.shape[0] - 1
    Jp, Jl, r, e, w, valid = dense_homogeneous(n_cams, lm_off, cam_idx, obs, cams, lms_h, norm, huber)
    sigma = 1.0 / (eps + np.sqrt((Jp * Jp).sum(0)))
    jl_scale = 1.0 / (eps + np.sqrt((Jl * Jl).sum(0)))
    Jps, Jls = Jp * sigma, Jl * jl_scale
    Nc = np.zeros((12 * n_cams, 11 * n_cams))
    for c in range(n_cams):
        Nc[12 * c : 12 * c + 12, 11 * c : 11 * c + 11] = scipy.linalg.null_space(cams[c][None, :])
    Nl = np.zeros((4 * n_l, 3 * n_l))
    for l in range(n_l):
        Nl[4 * l : 4 * l + 4, 3 * l : 3 * l + 3] = scipy.linalg.null_space(lms_h[l][None, :])
    Jpt, Jlt = Jps @ Nc, Jls @ Nl
    Hll = Jlt.T @ Jlt + lam * np.eye(3 * n_l)
    Hll_inv = np.zeros_like(Hll)
    for l in range(n_l):
        s = slice(3 * l, 3 * l + 3)
        Hll_inv[s, s] = np.linalg.inv(Hll[s, s])
    Hpl = Jpt.T @ Jlt
    E0 = Hpl @ Hll_inv @ Hpl.T
    B = Jpt.T @ Jpt + lam * np.eye(11 * n_cams)
    b = Jpt.T @ (r - Jlt @ (Hll_inv @ (Jlt.T @ r)))
    return dict(B=B, E0=E0, b=b, Nc=Nc, sigma=sigma)


# --------------------------------------------------------------------------- explicit SC (LinearizorSC)

def block_jacobi_inverse(S, dim):
    """Schur-Jacobi preconditioner: inverse of the dim x dim diagonal blocks of S
    (cg/preconditioner.hpp:66-118 as used by solver/linearizor_sc.cpp:129-135)."""
    M = np.zeros_like(S)
    for c in range(S.shape[0] // dim):
        s = slice(dim * c, dim * c + dim)
        M[s, s] = np.linalg.inv(S[s, s])
    return M


def pcg(S, b, M, min_iterations=0, max_iterations=500, eta=1e-2, reset_period=10):
    """Ceres-style PCG of cg/conjugate_gradient.hpp:112-290 as driven by
    solver/linearizor_base.cpp:104-125 (r_tolerance = -1, q_tolerance = eta; the result is negated).
    Returns (inc, num_iterations, status, iterates) with status 0 no convergence / 1 success /
    2 failure and iterates[k] = x after iteration k+1 (before the final negation)."""
    n = b.shape[0]
    x = np.zeros(n)
    iterates = []
    if np.linalg.norm(b) == 0.0:
        return -x, 0, 1, iterates
    r = b - S @ x
    rho, q0 = 1.0, -float(x @ (b + r))
    status, it = 0, 0
    p = None
    while True:
        it += 1
        z = M @ r if M is not None else r.copy()
        last_rho, rho = rho, float(r @ z)
        if rho == 0.0 or np.isinf(rho):
            status = 2
            break
        if it == 1:
            p = z
        else:
            beta = rho / last_rho
            if beta == 0.0 or np.isinf(beta):
                status = 2
                break
            p = z + beta * p
        q = S @ p
        pq = float(p @ q)
        if pq <= 0 or np.isinf(pq):
            status = 0
            break
        alpha = rho / pq
        if np.isinf(alpha):
            status = 2
            break
        x = x + alpha * p
        r = b - S @ x if it % reset_period == 0 else r - alpha * q
        iterates.append(x.copy())
        q1 = -float(x @ (b + r))
        zeta = it * (q1 - q0) / q1
        if zeta < eta and it >= min_iterations:
            status = 1
            break
        q0 = q1
        if it >= max_iterations:
            break
    return -x, it, status, np.array(iterates)
