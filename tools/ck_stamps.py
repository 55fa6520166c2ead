"""Where the time of e0_ck goes, phase by phase: in-kernel s_memtime stamps of a diagnostic build of the library
(tools/variants/ck_stamps.patch applied to a copy of the sources; the shipped library carries no stamp code).

    make -C povar_amd/csrc stamps && POVAR_LIB=build/libpovar_hip_stamps.so python tools/ck_stamps.py venice-1778 --variant 1
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("problem", nargs="?", default="venice-1778")
ap.add_argument("--variant", type=int, default=1)
a = ap.parse_args()
p = synth.make_bal_problem(a.problem)
ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
ctx.layout_finalize(True)
ctx.set_cameras(p.cams)
ctx.init_landmarks_pose(0.01)
assert ctx.linearize_pose(0.01)
ctx.prepare_pose(1e-4)
ctx.set_e0_kernel(a.variant)
li = ctx.layout_info()
NS = 40
buf = np.zeros(li.grid * 16 * NS, dtype=np.uint64)
assert ctx.L.povar_debug_ck_stamps(ctx.h, C.c_void_p(buf.ctypes.data), C.c_int64(buf.size)) == 0
for _ in range(3):
    ctx.power_series_pose(20)
ctx.synchronize()
n = ctx.L.povar_debug_ck_stamps(ctx.h, C.c_void_p(buf.ctypes.data), C.c_int64(buf.size))
assert n == buf.size, n
s = buf.reshape(li.grid, 16, NS).astype(np.float64)
nw = int((s[:, :, 0] > 0).any(axis=0).sum())
nb = li.ck_batches
names = ["start", "h in LDS", "gather issued", "barrier1", "fwd done", "barrier2", "Gu+barrier3", "bwd done"]
t0 = s[:, :nw, 0].min(axis=1, keepdims=True)  # the workgroup's first wavefront to start
print(f"cycles since the workgroup's start, median over {li.grid} workgroups; columns: wavefronts 0..{nw - 1}")
for b in range(nb):
    for i, nm in enumerate(names):
        v = np.median(s[:, :nw, 8 * b + i] - t0, axis=0)
        print(f"b{b} {nm:14s}" + "".join(f"{x / 1000:7.1f}" for x in v))
for i, nm in ((8 * nb, "last barrier"), (8 * nb + 1, "flushed")):
    v = np.median(s[:, :nw, i] - t0, axis=0)
    print(f"   {nm:14s}" + "".join(f"{x / 1000:7.1f}" for x in v))
for i, nm in ((20, 'b0 fwd tile0 done'), (21, 'b0 tile1 issued'), (22, 'b0 fwd tile1 done')):
    v = np.median(s[:, :nw, i] - t0, axis=0)
    print(f"   {nm:14s}" + "".join(f"{x / 1000:7.1f}" for x in v))
# Who does a barrier wait for?  Per workgroup: the wavefront that reaches "fwd done" / "bwd done" LAST, how far behind the
# workgroup's median wavefront it is, whether it walked a second tile (stamp 22 set), and what is left between its arrival
# and the barrier opening (the drain of its LDS traffic + the barrier itself)
for b in range(nb):
    for i_done, i_bar, nm in ((8 * b + 4, 8 * b + 5, "forward"), (8 * b + 7, 8 * b + 8 if b + 1 < nb else 8 * nb, "backward")):
        done = s[:, :nw, i_done] - t0
        bar = np.median(s[:, :nw, i_bar] - t0, axis=1)
        last = done.argmax(axis=1)
        lag = done.max(axis=1) - np.median(done, axis=1)
        two = (s[np.arange(li.grid), last, 22] > 0) if b == 0 and nm == "forward" else None
        hist = np.bincount(last, minlength=nw)
        print(f"b{b} {nm}: last wavefront by index " + " ".join(str(x) for x in hist) +
              f"; it is {np.median(lag) / 1000:.1f} k cycles behind the workgroup's median wavefront (90 %: {np.percentile(lag, 90) / 1000:.1f});"
              f" last 'done' -> barrier open {np.median(bar - done.max(axis=1)) / 1000:.1f} k" +
              (f"; the last one walked a second tile in {int(two.sum())} of {li.grid} workgroups" if two is not None else ""))
# Round 6: what happens between a wavefront's "fwd done" and the barrier behind the way forward opening (the
# 4.3 - 4.6 k cycles between the LAST wavefront's arrival and the opening).  Stamps 24 + 4 b .. 26 + 4 b: G and the way back's
# metadata requested / its first rows requested / s_waitcnt lgkmcnt(0) passed (LDS atomics and scalar loads drained).
for b in range(min(nb, 4)):
    if not (s[:, :nw, 24 + 4 * b] > 0).any():
        continue
    base = s[:, :nw, 8 * b + 4]
    names2 = ((24 + 4 * b, "G + metadata requested"), (25 + 4 * b, "first rows back requested"), (26 + 4 * b, "lgkmcnt(0) passed"),
              (8 * b + 5, "barrier open"))
    print(f"b{b}: cycles after the wavefront's own 'fwd done' (median over workgroups; columns: wavefronts)")
    for i, nm in names2:
        v = np.median(s[:, :nw, i] - base, axis=0)
        print(f"   {nm:26s}" + "".join(f"{x / 1000:7.2f}" for x in v))
    # the LAST wavefront of each workgroup (the one the barrier waits for): its own intervals
    last = base.argmax(axis=1)
    rows = np.arange(li.grid)
    prev = base[rows, last]
    parts = []
    for i, nm in names2:
        cur = s[rows, last, i]
        parts.append(f"{nm} +{np.median(cur - prev) / 1000:.2f} k")
        prev = cur
    print(f"   the last wavefront to finish its rows: " + ", ".join(parts))
# the kernel ends with its slowest workgroup: duration (first wavefront's start -> flushed) over the workgroups
dur = s[:, :nw, 8 * nb + 1].max(axis=1) - s[:, :nw, 0].min(axis=1)
order = np.argsort(dur)
print("workgroup durations (k cycles): min %.1f  10 %% %.1f  median %.1f  90 %% %.1f  max %.1f;  slowest workgroups: %s" % (
    dur.min() / 1000, np.percentile(dur, 10) / 1000, np.median(dur) / 1000, np.percentile(dur, 90) / 1000, dur.max() / 1000,
    " ".join(f"{int(w)}:{dur[w] / 1000:.1f}" for w in order[-6:][::-1])))
ctx.close()
