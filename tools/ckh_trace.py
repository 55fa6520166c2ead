"""Runs the STEP-2 term loop (solve_joint's power series) of one problem with e0_lpl_h (0) and e0_ck_h (1) in turn, a few
solves each: the workload of `rocprofv3 --kernel-trace --stats` / `--pmc` runs that compare the two kernels; prints the
relative difference of the 20-term increments.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 tools/ckh_trace.py venice-1778 --variants 0,1
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("problem", nargs="?", default="venice-1778")
ap.add_argument("--variants", default="0,1")
ap.add_argument("--solves", type=int, default=3)
ap.add_argument("--robust", default="NONE")
ap.add_argument("--popularity", default="zipf1")
a = ap.parse_args()
p = synth.make_bal_problem(a.problem, a.popularity)
ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=a.robust, e0_mode=capi.E0_IMPLICIT_LDSACC)
ctx.layout_finalize(True)
ctx.set_cameras(p.cams)
ctx.init_landmarks_pose(0.01)
assert ctx.linearize_pose(0.01)
ctx.prepare_pose(1e-4, capi.POWER_VARPROJ)
ctx.normalize_joint()           # the step-2 system at the state step 1 starts from (as bench.py --step 2)
assert ctx.linearize_homogeneous()
ctx.prepare_joint(1e-4)
incs = {}
for v in [int(x) for x in a.variants.split(",")]:
    ctx.set_e0_kernel(v)
    for _ in range(a.solves):
        ctx.power_series_pose(20)
    ctx.synchronize()
    incs[v] = ctx.get_increment(11)
li = ctx.layout_info()
print({"problem": a.problem, "robust": a.robust, "ckh_batches": li.ckh_batches, "ckh_slots": li.ckh_slots, "ckh_chunks": li.ckh_chunks,
       "ckh_cold_chunks": li.ckh_cold_chunks,
       "inc_rel_diff": float(np.linalg.norm(incs[1] - incs[0]) / np.linalg.norm(incs[0])) if 0 in incs and 1 in incs else None})
ctx.close()
