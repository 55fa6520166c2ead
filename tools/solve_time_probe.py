#!/usr/bin/env python3
"""Wall time of every inner solve of an LM-like call sequence on one context (what `bal` reports as
solve_reduced_system_time), with the state of the row placement beside it: where the one-off costs of a context land
(kernel timing, graph capture, the swap to the placed rows).  usage: solve_time_probe.py [shape] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "venice-1778"
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 12
p = synth.make_bal_problem(shape, init="gt", init_noise=0.02)
t0 = time.perf_counter()
ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
print(f"create {1e3 * (time.perf_counter() - t0):.1f} ms")
ctx.set_cameras(p.cams)
ctx.init_landmarks_pose(0.01)
lam = 1e-4
for it in range(n_it):
    t = time.perf_counter()
    assert ctx.linearize_pose(0.01)
    t_lin = time.perf_counter() - t
    t = time.perf_counter()
    ctx.prepare_pose(lam)
    ctx.synchronize()
    t_prep = time.perf_counter() - t
    t = time.perf_counter()
    its, st = ctx.power_series_pose(20, 0.0, -1.0)
    inc = ctx.get_increment()
    t_solve = time.perf_counter() - t
    li = ctx.layout_info()
    print(f"step 1 it {it}: linearize {1e3 * t_lin:6.2f} prepare {1e3 * t_prep:6.2f} solve {1e3 * t_solve:6.2f} ms ({1e6 * t_solve / 20:6.1f} us/term) "
          f"placement {li.placement} e0_kernel {li.e0_kernel} auto {li.e0_auto} since create {time.perf_counter() - t0:5.2f} s")
    ctx.apply_pose(capi.POWER_VARPROJ, 0.01, inc)
    lam *= 0.5
ctx.normalize_joint()
for it in range(4):
    t = time.perf_counter()
    assert ctx.linearize_homogeneous()
    t_lin = time.perf_counter() - t
    t = time.perf_counter()
    ctx.prepare_joint(lam)
    ctx.synchronize()
    t_prep = time.perf_counter() - t
    t = time.perf_counter()
    inc, its, st, rc = ctx.solve_joint(lam, 20)
    t_solve = time.perf_counter() - t
    li = ctx.layout_info()
    print(f"step 2 it {it}: linearize {1e3 * t_lin:6.2f} prepare {1e3 * t_prep:6.2f} solve (incl. prepare) {1e3 * t_solve:6.2f} ms placement {li.placement} "
          f"e0_kernel_h {li.e0_kernel_h} auto {li.e0_auto_h}")
    ctx.apply_joint(inc)
    ctx.normalize_joint()
ctx.close()
