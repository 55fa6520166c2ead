#!/usr/bin/env python3
"""Calls povar_error_pose (lpl_pass<1>) in a loop on the venice-1778 shape: a target for rocprofv3 passes of the
single-pass lane-per-landmark kernel.  usage: error_loop.py [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = synth.make_bal_problem("venice-1778")
ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
ctx.layout_finalize()  # steady state: the placed rows, not the natural order povar_create starts on
ctx.set_cameras(p.cams)
ctx.init_landmarks_pose(0.01)
for _ in range(3):
    ctx.error_pose(0.01)
ctx.synchronize()
t = time.perf_counter()
for _ in range(n):
    ctx.error_pose(0.01)
ctx.synchronize()
print("error_pose %.1f us per call (wall)" % ((time.perf_counter() - t) / n * 1e6))
