#!/usr/bin/env python3
"""Per-term time of the two forms of the power series (per-term kernels in a hipGraph against the resident kernel
series_res) on one context: a BASELINE shape, or rank 0's landmark shard of it at world = N.
usage: res_term_time.py shape [world] [robust]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "trafalgar-257"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    robust = sys.argv[3] if len(sys.argv) > 3 else "NONE"
    p = synth.make_bal_problem(shape)
    lb, le = capi.shard_range(p.lm_off, world, 0)
    ob, oe = int(p.lm_off[lb]), int(p.lm_off[le])
    ctx = capi.Context(p.n_cams, p.lm_off[lb:le + 1] - p.lm_off[lb], p.cam_idx[ob:oe], p.obs[ob:oe], robust_norm=robust,
                       e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.layout_finalize(True)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(0.01)
    assert ctx.linearize_pose(0.01)
    ctx.prepare_pose(1e-4)
    li = ctx.layout_info()
    m, k = 20, 50
    out = {}
    for mode in (0, 1):
        if mode == 1 and not li.res_ready:
            continue
        ctx.set_series_kernel(mode)
        for _ in range(3):
            ctx.power_series_pose(m, 0.0, -1.0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            ctx.power_series_pose(m, 0.0, -1.0)
        ctx.synchronize()
        out[mode] = (time.perf_counter() - t0) / (k * m) * 1e6
        inc = ctx.get_increment()
        out[("inc", mode)] = inc
    li = ctx.layout_info()
    import numpy as np
    d = np.linalg.norm(out[("inc", 1)] - out[("inc", 0)]) / np.linalg.norm(out[("inc", 0)]) if 1 in out else float("nan")
    print(f"{shape} world={world} {robust}: {oe - ob} obs, {le - lb} lms; per-term kernels {out[0]:.2f} us/term"
          + (f", resident {out[1]:.2f} us/term (|d inc| {d:.1e}; {li.res_wgs} wgs x {li.res_waves} waves, {li.res_rounds} x {li.res_rows} rows, "
             f"{li.res_records} records, max {li.res_max_chunks} chunks / {li.res_max_cams} cams / {li.res_max_lms} lms / {li.res_max_oq} owned records, "
             f"order {li.res_order}, lds {li.res_lds_bytes}, build {li.res_build_ms:.0f} ms, failed {li.res_failed})" if 1 in out else ", no resident layout"))
    ctx.close()


if __name__ == "__main__":
    main()
