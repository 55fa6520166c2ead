#!/usr/bin/env python3
"""Generates the committed golden fixtures from the independent NumPy restatement
(oracle/povar_numpy.py).  Run in the build container only:  python tests/golden/make_golden.py

The reference ships no golden vectors for this path and cannot be run here, so
these fixtures pin the C oracle and the HIP path against a second, dense implementation.
Every array is fp64; each file stays below ~100 KB.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import povar_numpy as N  # noqa: E402
from povar_amd import synth  # noqa: E402

ALPHA, LAM, M, EPS = 0.01, 1e-4, 20, 1e-5


def storage_from_dense(p, s1):
    n_o = p.n_obs
    lm_of = np.repeat(np.arange(p.n_lms), np.diff(p.lm_off))
    st = np.zeros((4 * n_o, 16))
    for i in range(n_o):
        c, l = p.cam_idx[i], lm_of[i]
        st[4 * i : 4 * i + 4, :12] = s1["Jps"][4 * i : 4 * i + 4, 12 * c : 12 * c + 12]
        st[4 * i : 4 * i + 4, 12:15] = s1["Jls"][4 * i : 4 * i + 4, 3 * l : 3 * l + 3]
        st[4 * i : 4 * i + 4, 15] = s1["r"][4 * i : 4 * i + 4]
    return st


def step1_fixture(p, norm, huber, with_storage):
    lms = N.init_landmarks_pose(ALPHA, p.lm_off, p.cam_idx, p.obs, p.cams)
    s1 = N.step1(ALPHA, p.n_cams, p.lm_off, p.cam_idx, p.obs, p.cams, lms, LAM, M, EPS, norm, huber)
    cams_new, lms_new, l_diff = N.apply_varproj(ALPHA, p.n_cams, p.lm_off, p.cam_idx, p.obs, p.cams, lms, s1, s1["inc"]) \
        if norm == "NONE" else (None, None, None)
    s1p = N.step1(ALPHA, p.n_cams, p.lm_off, p.cam_idx, p.obs, p.cams, lms, LAM, M, EPS, norm, huber, lam_lm=LAM)
    cams_p, lms_p, l_diff_p = N.apply_poba(p.n_cams, p.lm_off, p.cams, lms, s1p, s1p["inc"], LAM)
    out = dict(n_cams=p.n_cams, lm_off=p.lm_off, cam_idx=p.cam_idx, obs=p.obs, cams=p.cams, lms=lms,
               alpha=ALPHA, lam=LAM, m=M, eps=EPS, huber=huber,
               diag2=s1["diag2"], sigma=s1["sigma"], jl_scale=s1["jl_scale"], hll_inv=s1["hll_inv"],
               b=s1["b"], terms=s1["terms"], inc=s1["inc"], cost=s1["cost"], rho=s1["rho"],
               poba_inc=s1p["inc"], poba_lms_new=lms_p, poba_l_diff=l_diff_p, poba_cams_new=cams_p)
    if norm == "NONE":
        out.update(varproj_cams_new=cams_new, varproj_lms_new=lms_new, varproj_l_diff=l_diff)
    if with_storage:
        out.update(storage=storage_from_dense(p, s1), b_inv=s1["b_inv"])
    return out


def step2_fixture(p):
    rng = np.random.default_rng(11)
    cams = rng.normal(size=(p.n_cams, 12))
    cams[:, 8:11] *= 0.1
    cams[:, 11] = 5 + rng.random(p.n_cams)
    cams /= np.linalg.norm(cams, axis=1, keepdims=True)
    lms_h = np.concatenate([rng.normal(size=(p.n_lms, 3)), np.ones((p.n_lms, 1))], 1)
    obs = p.obs / 500.0
    s2 = N.step2(p.n_cams, p.lm_off, p.cam_idx, obs, cams, lms_h, LAM, 10, EPS)
    return dict(n_cams=p.n_cams, lm_off=p.lm_off, cam_idx=p.cam_idx, obs=obs, cams=cams, lms_h=lms_h,
                lam=LAM, m=10, eps=EPS, diag2=s2["diag2"], sigma=s2["sigma"], jl_scale=s2["jl_scale"],
                term_norms=s2["term_norms"], ambient_terms=s2["ambient_terms"], ambient_inc=s2["ambient_inc"],
                l_diff=s2["l_diff"], cams_new=s2["cams_new"], lms_new=s2["lms_new"],
                cams_norm=s2["cams_norm"], lms_norm=s2["lms_norm"], cost=s2["cost"])


def sc_fixture(p):
    """Explicit-Schur-complement solvers (LinearizorSC): dense S = B - E0, Schur-Jacobi PCG iterates and
    the exact solve, step 1 (12 n_cams) and step 2 (ambient coordinates: basis independent)."""
    lms = N.init_landmarks_pose(ALPHA, p.lm_off, p.cam_idx, p.obs, p.cams)
    s1 = N.step1(ALPHA, p.n_cams, p.lm_off, p.cam_idx, p.obs, p.cams, lms, LAM, 1, EPS)
    S = s1["B"] - s1["E0"]
    M = N.block_jacobi_inverse(S, 12)
    inc, it, status, iterates = N.pcg(S, s1["b"], M, eta=1e-2)
    inc8, it8, status8, iterates8 = N.pcg(S, s1["b"], M, eta=0.0, max_iterations=12)
    out = dict(n_cams=p.n_cams, lm_off=p.lm_off, cam_idx=p.cam_idx, obs=p.obs, cams=p.cams, lms=lms,
               alpha=ALPHA, lam=LAM, eps=EPS, S_diag_blocks=np.array([S[12 * c:12 * c + 12, 12 * c:12 * c + 12]
                                                                      for c in range(p.n_cams)]),
               S_row0=S[:12], b=s1["b"], pcg_inc=inc, pcg_iterations=it, pcg_status=status,
               pcg_iterates_eta0=iterates8, exact=s1["exact"])
    # step 2
    g2 = step2_fixture(p)
    s2 = N.step2_system(p.n_cams, p.lm_off, p.cam_idx, g2["obs"], g2["cams"], g2["lms_h"], LAM, EPS)
    S2 = s2["B"] - s2["E0"]
    M2 = N.block_jacobi_inverse(S2, 11)
    inc2, it2, status2, iterates2 = N.pcg(S2, s2["b"], M2, eta=1e-2)
    _, _, _, iterates2_eta0 = N.pcg(S2, s2["b"], M2, eta=0.0, max_iterations=6)
    out.update(obs2=g2["obs"], cams2=g2["cams"], lms_h2=g2["lms_h"],
               ripcg_ambient_inc=s2["Nc"] @ inc2, ripcg_iterations=it2, ripcg_status=status2,
               ripcg_ambient_iterates_eta0=iterates2_eta0 @ s2["Nc"].T,
               ripcg_ambient_exact=s2["Nc"] @ np.linalg.solve(S2, -s2["b"]))
    return out


def main():
    small = synth.make_problem(6, 40, 150, seed=3)
    medium = synth.make_problem(49, 300, 1230, seed=49)
    np.savez_compressed(os.path.join(HERE, "step1_small_none.npz"), **step1_fixture(small, "NONE", 1.0, True))
    np.savez_compressed(os.path.join(HERE, "step1_small_huber.npz"), **step1_fixture(small, "HUBER", 30.0, False))
    np.savez_compressed(os.path.join(HERE, "step1_medium_none.npz"), **step1_fixture(medium, "NONE", 1.0, False))
    np.savez_compressed(os.path.join(HERE, "step2_small.npz"), **step2_fixture(small))
    np.savez_compressed(os.path.join(HERE, "sc_small.npz"), **sc_fixture(small))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
