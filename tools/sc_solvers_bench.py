#!/usr/bin/env python3
"""Times the explicit-Schur-complement solvers (PCG / CHOLESKY / RIPCG) on a synthetic BAL shape and checks
them through size-independent properties: |S x + b| / |b| with S applied by the independent
right_mul_e0 entry point, PCG(tight eta) == CHOLESKY.   usage: sc_solvers_bench.py [shape] [lambda]"""
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
    lam = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
    p = synth.make_bal_problem(shape)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.layout_finalize()
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(0.01)
    ctx.set_jl_col_scaling(False)
    assert ctx.linearize_pose(0.01)

    def residual(x):
        bm = ctx.get_buffer(capi.BUF_SC_BLOCKDIAG).reshape(p.n_cams, 12, 12)
        b = ctx.get_buffer(capi.BUF_B)
        Sx = np.einsum("cij,cj->ci", bm, x.reshape(-1, 12)).ravel() - ctx.right_mul_e0_pose(x)
        return np.linalg.norm(Sx + b) / np.linalg.norm(b)

    out = {}
    for eta, mx in [(1e-2, 500), (1e-6, 500)]:
        ctx.solve_pose_sc(lam, capi.SC_PCG, 0, mx, eta)  # warm
        t0 = time.perf_counter()
        inc, it, st, rc = ctx.solve_pose_sc(lam, capi.SC_PCG, 0, mx, eta)
        dt = time.perf_counter() - t0
        out[eta] = inc
        print(f"PCG eta={eta:g}: {it} iterations, status {st}, {dt*1e3:.1f} ms total, {dt/it*1e6:.0f} us/iteration, "
              f"|Sx+b|/|b| = {residual(inc):.3e}", flush=True)
    t0 = time.perf_counter()
    inc_c, it, st, rc = ctx.solve_pose_sc(lam, capi.SC_CHOLESKY)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    inc_c, it, st, rc = ctx.solve_pose_sc(lam, capi.SC_CHOLESKY)
    dt = time.perf_counter() - t0
    n = 12 * p.n_cams
    print(f"CHOLESKY n={n}: first {dt1*1e3:.1f} ms, then {dt*1e3:.1f} ms ({n**3/3/dt/1e12:.2f} TFLOP/s incl. assembly), rc {rc}, "
          f"|Sx+b|/|b| = {residual(inc_c):.3e}", flush=True)
    print("PCG(1e-6) vs CHOLESKY:", np.linalg.norm(out[1e-6] - inc_c) / np.linalg.norm(inc_c))
    print("device MiB:", ctx.device_bytes() >> 20)
    ctx.close()


if __name__ == "__main__":
    main()
