#!/usr/bin/env python3
"""Term time of the venice-1778 shape with a tail of LONG landmarks (> 64 observations, handled by the
lm_long driver): `n_long` landmarks are made by merging `merge` consecutive synthetic landmarks each
(duplicate cameras dropped).  Real BAL photo collections have such tracks; the seeded synthetic shapes do not.
usage: long_track_bench.py [n_long] [merge]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402


def main():
    n_long = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    merge = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    p = synth.make_bal_problem("venice-1778")
    k = np.diff(p.lm_off)
    lm_of = np.repeat(np.arange(p.n_lms), k)
    # landmarks [0, n_long * merge) are merged in groups of `merge`
    grp = np.where(lm_of < n_long * merge, lm_of // merge, lm_of - n_long * merge + n_long)
    order = np.lexsort((p.cam_idx, grp))
    grp, cam, obs = grp[order], p.cam_idx[order], p.obs[order]
    keep = np.ones(len(grp), dtype=bool)
    keep[1:] = (grp[1:] != grp[:-1]) | (cam[1:] != cam[:-1])
    grp, cam, obs = grp[keep], cam[keep], obs[keep]
    n_l = int(grp.max()) + 1
    lm_off = np.zeros(n_l + 1, dtype=np.int64)
    np.cumsum(np.bincount(grp, minlength=n_l), out=lm_off[1:])
    deg = np.diff(lm_off)
    print(f"{n_l} landmarks, {len(cam)} observations, {int((deg > 64).sum())} long landmarks (max {deg.max()}, "
          f"{int(deg[deg > 64].sum())} observations on them)", flush=True)
    for mode in (capi.E0_IMPLICIT_LDSACC, capi.E0_IMPLICIT):
        ctx = capi.Context(p.n_cams, lm_off.astype(np.int32), cam.astype(np.int32), obs, e0_mode=mode)
        ctx.layout_finalize()
        ctx.set_cameras(p.cams)
        ctx.init_landmarks_pose(0.01)
        assert ctx.linearize_pose(0.01)
        ctx.prepare_pose(1e-4)
        m = 20
        for _ in range(3):
            ctx.power_series_pose(m, 0.0, -1.0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.power_series_pose(m, 0.0, -1.0)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        inc = ctx.get_increment()
        print(f"mode {mode}: {dt / (10 * m) * 1e6:.1f} us per term ({10 * m / dt:.0f} terms/s), |inc| = {np.linalg.norm(inc):.6e}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
