#!/usr/bin/env python3
"""Host wall time of povar_create (layout construction + uploads) per phase: POVAR_LAYOUT_TIMING=1 prints the phases of
povar_create and build_lpl on stderr; povar_get_layout_info carries the total.   usage: create_time.py [problem ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("POVAR_LAYOUT_TIMING", "1")
from povar_amd import capi, synth  # noqa: E402

for name in (sys.argv[1:] or ["venice-1778"]):
    p = synth.make_bal_problem(name)
    for rep in range(2):
        t0 = time.perf_counter()
        ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
        dt = time.perf_counter() - t0
        print(f"{name}: povar_create {ctx.layout_info().create_ms:.1f} ms (python wall {dt * 1e3:.1f} ms), host threads {os.cpu_count()}", flush=True)
        t1 = time.perf_counter()
        ctx.layout_finalize(wait=True)
        li = ctx.layout_info()
        print(f"{name}: row placement {['none', 'inside povar_create', 'pending', 'on a host thread'][li.placement]}: "
              f"{li.placement_ms:.1f} ms of host wall time in the background, waited {(time.perf_counter() - t1) * 1e3:.1f} ms more after povar_create", flush=True)
        ctx.close()
