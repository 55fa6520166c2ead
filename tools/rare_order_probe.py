#!/usr/bin/env python3
"""What would the lane-per-landmark / camera-chunk layouts gain from workgroups cut out of a landmark order by RAREST
camera (the resident layout's order, res_layout.hpp) on a graph without locality?  The problem's landmarks are permuted
outside the library and the existing `range` strategy is forced; compared with the library's own choice on the same
(permuted) problem.  usage: rare_order_probe.py [shape] [popularity]"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(tag, env):
    code = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %r)
from povar_amd import capi, synth
shape, pop = %r, %r
p = synth.make_bal_problem(shape, popularity=pop)
cnt = np.bincount(p.cam_idx, minlength=p.n_cams)
rank = np.empty(p.n_cams, dtype=np.int64); rank[np.argsort(-cnt, kind="stable")] = np.arange(p.n_cams)
lm_of_obs = np.repeat(np.arange(p.n_lms), np.diff(p.lm_off))
r = rank[p.cam_idx]
r1 = np.zeros(p.n_lms, dtype=np.int64); np.maximum.at(r1, lm_of_obs, r)
order = np.argsort(r1, kind="stable")
k = np.diff(p.lm_off)[order]
lm_off = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
idx = np.concatenate([np.arange(p.lm_off[l], p.lm_off[l + 1]) for l in order]) if os.environ.get("PROBE_PERMUTE") == "1" else np.arange(p.n_obs)
if os.environ.get("PROBE_PERMUTE") != "1":
    lm_off = p.lm_off
ctx = capi.Context(p.n_cams, lm_off, p.cam_idx[idx], p.obs[idx], e0_mode=capi.E0_IMPLICIT_LDSACC)
ctx.layout_finalize(True)
ctx.set_cameras(p.cams); ctx.init_landmarks_pose(0.01); assert ctx.linearize_pose(0.01); ctx.prepare_pose(1e-4)
for _ in range(3): ctx.power_series_pose(20, 0.0, -1.0)
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(40): ctx.power_series_pose(20, 0.0, -1.0)
ctx.synchronize(); dt = time.perf_counter() - t0
li = ctx.layout_info()
TAG = %r
print(f"{TAG}: {800 / dt:8.0f} terms/s  strategy {li.strategy} resident {1 - li.n_cold / li.n_obs:.4f} e0_kernel {li.e0_kernel} tune lpl {li.tune_lpl_us:.1f} ck {li.tune_ck_us:.1f} "
      f"chunks {li.ck_chunks} own-record chunks {li.ck_cold_chunks} batches {li.ck_batches} tiles_max {li.ck_tiles_max} records {li.ck_part_rec}")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), shape, pop, tag)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-800:]])[-1])


shape = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
pop = sys.argv[2] if len(sys.argv) > 2 else "zipf1"
run("file order, own choice", {})
run("file order, ranges forced", {"POVAR_LPL_STRATEGY": "range"})
run("rarest-camera order, ranges forced", {"PROBE_PERMUTE": "1", "POVAR_LPL_STRATEGY": "range"})
run("rarest-camera order, own choice", {"PROBE_PERMUTE": "1"})
