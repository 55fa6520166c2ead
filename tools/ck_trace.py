"""Runs the term loop of one problem with every e0_ck instantiation in turn (and e0_lpl), a few solves each: the
workload of `rocprofv3 --kernel-trace --stats` / `--pmc` runs that compare the instantiations kernel by kernel.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 tools/ck_trace.py venice-1778 --variants 0,1,2
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("problem", nargs="?", default="venice-1778")
ap.add_argument("--variants", default="0,1,2,3,4,5,6")
ap.add_argument("--solves", type=int, default=3)
ap.add_argument("--robust", default="NONE")
a = ap.parse_args()
p = synth.make_bal_problem(a.problem)
ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=a.robust, e0_mode=capi.E0_IMPLICIT_LDSACC)
ctx.layout_finalize(True)
ctx.set_cameras(p.cams)
ctx.init_landmarks_pose(0.01)
assert ctx.linearize_pose(0.01)
ctx.prepare_pose(1e-4)
for v in [int(x) for x in a.variants.split(",")]:
    ctx.set_e0_kernel(v)
    for _ in range(a.solves):
        ctx.power_series_pose(20)
    ctx.synchronize()
ctx.close()
