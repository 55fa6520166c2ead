"""POVAR_DETERMINISTIC=1 against the default mode on one problem: time per power-series term of a 20-term solve (replayed
graph), bit identity of repeated solves, and the distance of increment / last term from the default mode's.
usage: python3 tools/det_probe.py [problem=venice-1778] [robust=NONE] [solves=20]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
robust = sys.argv[2] if len(sys.argv) > 2 else "NONE"
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 20
p = synth.make_bal_problem(name)


def run(env):
    for k in ("POVAR_DETERMINISTIC", "POVAR_DET_CK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=robust, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.layout_finalize(True)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(0.01)
    t0 = time.perf_counter()
    assert ctx.linearize_pose(0.01)
    ctx.prepare_pose(1e-4, capi.POWER_VARPROJ)
    ctx.synchronize()
    t_lin = time.perf_counter() - t0
    incs = [ctx.solve_pose(1e-4, capi.POWER_VARPROJ, 20)[0] for _ in range(3)]
    ctx.prepare_pose(1e-4, capi.POWER_VARPROJ)
    ctx.power_series_pose(20)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(solves):
        ctx.power_series_pose(20)
    ctx.synchronize()
    us = (time.perf_counter() - t0) / solves / 20 * 1e6
    term = ctx.get_term()
    li = ctx.layout_info()
    ctx.close()
    return incs, term, us, t_lin, li.e0_kernel


rows = []
for tag, env in (("default", {}), ("deterministic, fixed-point e0_ck", {"POVAR_DETERMINISTIC": "1"}),
                 ("deterministic, gather form", {"POVAR_DETERMINISTIC": "1", "POVAR_DET_CK": "0"})):
    incs, term, us, t_lin, kern = run(env)
    rows.append((tag, incs, term))
    same = all(np.array_equal(incs[0], x) for x in incs[1:])
    print(f"{name} {robust} {tag}: {us:.1f} us per term (host clock, {solves} solves of 20 terms), linearise + prepare {1e3 * t_lin:.2f} ms, "
          f"e0_kernel {kern}, repeated solves bit-identical: {same}", flush=True)
ref_inc, ref_term = rows[0][1][0], rows[0][2]
for tag, incs, term in rows[1:]:
    print(f"  {tag}: |inc - default| / |default| = {np.linalg.norm(incs[0] - ref_inc) / np.linalg.norm(ref_inc):.2e}, "
          f"last term {np.linalg.norm(term - ref_term) / np.linalg.norm(ref_term):.2e}")
print(f"  fixed-point vs gather: inc {np.linalg.norm(rows[1][1][0] - rows[2][1][0]) / np.linalg.norm(rows[2][1][0]):.2e}")
