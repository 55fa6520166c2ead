#!/usr/bin/env python3
"""One LM iteration's stages in a loop, through the C ABI: a target for rocprofv3 passes (kernel trace, PMC FETCH_SIZE /
WRITE_SIZE) over EVERY kernel of the path, not only the term pair.
    stage_loop.py <problem> [--step 1|2] [--robust-norm HUBER --huber 20] [--iters N] [--m M]
step 1: error_pose, linearize_pose, prepare_pose, solve (m terms), backup, apply_pose, error_pose, restore
step 2: the same with the homogeneous / joint entry points, from the normalised step-1 start."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("problem")
ap.add_argument("--step", type=int, default=1)
ap.add_argument("--robust-norm", default="NONE")
ap.add_argument("--huber", type=float, default=1.0)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--m", type=int, default=4)
ap.add_argument("--popularity", default="zipf1")
a = ap.parse_args()
alpha, lam = 0.01, 1e-4
p = synth.make_bal_problem(a.problem, a.popularity)
ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC, robust_norm=a.robust_norm, huber=a.huber)
ctx.layout_finalize()  # steady state: the placed rows, not the natural order povar_create starts on
ctx.set_cameras(p.cams)
ctx.init_landmarks_pose(alpha)
t0 = time.perf_counter()
if a.step == 1:
    for _ in range(a.iters):
        ctx.error_pose(alpha)
        assert ctx.linearize_pose(alpha)
        inc, it, st, rc = ctx.solve_pose(lam, capi.POWER_VARPROJ, a.m)
        ctx.backup_pose()
        ctx.apply_pose(capi.POWER_VARPROJ, alpha, inc)
        ctx.error_pose(alpha)
        ctx.restore_pose()
else:
    ctx.normalize_joint()
    for _ in range(a.iters):
        ctx.error_homogeneous()
        assert ctx.linearize_homogeneous()
        inc, it, st, rc = ctx.solve_joint(lam, a.m)
        ctx.backup_joint()
        ctx.apply_joint(inc)
        ctx.error_homogeneous()
        ctx.restore_joint()
ctx.synchronize()
print("%s step %d: %.2f ms per LM iteration (wall, m = %d)" % (a.problem, a.step, (time.perf_counter() - t0) / a.iters * 1e3, a.m))
