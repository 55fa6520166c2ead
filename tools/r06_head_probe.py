"""Round 6: the per-camera step at the head of e0_ck (POVAR_CK_HEAD=1) against the per-camera kernel as a launch of its own
(POVAR_CK_HEAD=0) on one problem: same 20-term increment, time per term of the replayed term loop.
    python tools/r06_head_probe.py [problem] [--robust HUBER]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("problem", nargs="?", default="venice-1778")
ap.add_argument("--robust", default="NONE")
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
p = synth.make_bal_problem(a.problem)
out = {}
for head in ("0", "1", "0", "1"):
    os.environ["POVAR_CK_HEAD"] = head
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=a.robust, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.layout_finalize(True)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(0.01)
    assert ctx.linearize_pose(0.01)
    ctx.prepare_pose(1e-4)
    ctx.set_e0_kernel(1)
    for _ in range(100):
        ctx.power_series_pose(20)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        ctx.power_series_pose(20)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    inc = ctx.get_increment()
    li = ctx.layout_info()
    print(f"POVAR_CK_HEAD={head}: {1e6 * dt / (a.reps * 20):.2f} us per term; head_ready {getattr(li, 'head_ready', '?')} active {getattr(li, 'head_active', '?')} failed {getattr(li, 'head_failed', '?')}; |inc| {np.linalg.norm(inc):.6e}")
    out.setdefault(head, inc)
    ctx.close()
r = np.linalg.norm(out["1"] - out["0"]) / np.linalg.norm(out["0"])
print(f"increment with heads against without: rel. diff {r:.2e}")
assert r < 1e-10
