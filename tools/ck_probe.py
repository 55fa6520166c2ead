"""e0_ck (lane = camera chunk, landmarks in LDS) against e0_lpl (lane = landmark, cameras in LDS) on one problem:
the same E0 x and the same 20-term increment to rounding, and the time of a term with each kernel.

    python tools/ck_probe.py [problem ...] [--robust HUBER] [--variants 1,2,3] [--reps 30]

Prints one JSON line per problem.  Measurement tool (GPU)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402

ALPHA, LAM, M = 0.01, 1e-4, 20


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def problem(name, popularity=None):
    if name == "local-900":
        return synth.make_problem(900, 40000, 200000, seed=9, popularity="local")
    if name.startswith("zipf-"):  # zipf-<cams>-<lms>-<obs>
        _, c, l, o = name.split("-")
        return synth.make_problem(int(c), int(l), int(o), seed=3)
    if popularity:
        return synth.make_bal_problem(name, popularity=popularity)
    return synth.make_bal_problem(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("problems", nargs="*", default=["trafalgar-257", "venice-1778"])
    ap.add_argument("--robust", default="NONE")
    ap.add_argument("--variants", default="1,2,3,4,5")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--popularity", default=None)
    a = ap.parse_args()
    variants = [int(v) for v in a.variants.split(",") if v]
    for name in a.problems:
        p = problem(name, a.popularity)
        ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=a.robust, e0_mode=capi.E0_IMPLICIT_LDSACC)
        ctx.layout_finalize(True)
        ctx.set_cameras(p.cams)
        ctx.init_landmarks_pose(ALPHA)
        assert ctx.linearize_pose(ALPHA)
        ctx.prepare_pose(LAM)
        li = ctx.layout_info()
        out = {"problem": name, "robust": a.robust, "n_obs": p.n_obs, "ck_ready": li.ck_ready, "ck_batches": li.ck_batches,
               "ck_slots": li.ck_slots, "ck_tiles_max": li.ck_tiles_max, "ck_chunks": li.ck_chunks, "ck_rows": li.ck_rows,
               "lpl_rows": li.n_rows, "ck_cold_chunks": li.ck_cold_chunks, "ck_part_rec": li.ck_part_rec,
               "ck_build_ms": round(li.ck_build_ms, 1), "lane_per_landmark": li.lane_per_landmark}
        x = np.random.default_rng(5).normal(size=12 * p.n_cams)

        def run(kernel):
            ctx.set_e0_kernel(kernel)
            y = ctx.right_mul_e0_pose(x)
            ctx.power_series_pose(M)
            inc = ctx.get_increment()
            for _ in range(3):
                ctx.power_series_pose(M)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                ctx.power_series_pose(M)
            ctx.synchronize()
            us = (time.perf_counter() - t0) / (a.reps * M) * 1e6
            return y, inc, us

        y0, inc0, us0 = run(0)
        out["lpl_us_per_term"] = round(us0, 2)
        if li.ck_ready:
            for v in variants:
                y, inc, us = run(v)
                out[f"ck{v}"] = {"us_per_term": round(us, 2), "e0_rel": rel(y, y0), "inc_rel": rel(inc, inc0)}
        print(json.dumps(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
