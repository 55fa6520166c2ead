#!/usr/bin/env python3
"""Per-term time of ONE rank's landmark shard of the venice-1778 shape on one GPU (the compute part of a
strong-scaling step at world = N; the all-reduce latency is not included: POVAR_FORCE_COMM=1 adds a 1-rank
communicator so that the kernel sequence is the sharded one).  usage: shard_term_time.py N [shape] [e0_mode 0-3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from povar_amd import capi, synth  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    shape = sys.argv[2] if len(sys.argv) > 2 else "venice-1778"
    p = synth.make_bal_problem(shape)
    lb, le = capi.shard_range(p.lm_off, world, 0)
    ob, oe = int(p.lm_off[lb]), int(p.lm_off[le])
    ctx = capi.Context(p.n_cams, p.lm_off[lb:le + 1] - p.lm_off[lb], p.cam_idx[ob:oe], p.obs[ob:oe],
                       e0_mode=int(sys.argv[3]) if len(sys.argv) > 3 else capi.E0_IMPLICIT_LDSACC)
    if os.environ.get("POVAR_FORCE_COMM"):
        ctx.comm_init(1, 0, capi.comm_unique_id())
        if os.environ.get("POVAR_P2P"):  # per-term exchange through the push/reduce kernels (world of one)
            ctx.p2p_attach(1, 0, [ctx.p2p_export(1)])
    ctx.layout_finalize()  # steady state: the placed rows
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(0.01)
    assert ctx.linearize_pose(0.01)
    ctx.prepare_pose(1e-4)
    m = 20
    for _ in range(3):
        ctx.power_series_pose(m, 0.0, -1.0)
    ctx.synchronize()
    t0 = time.perf_counter()
    k = 20
    for _ in range(k):
        ctx.power_series_pose(m, 0.0, -1.0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(True)
    for _ in range(5):
        ctx.power_series_pose(m, 0.0, -1.0)
    ctx.synchronize()
    pr = ctx.profile_get()
    print(f"{shape} world={world}: shard {le - lb} landmarks / {oe - ob} obs; {dt / (k * m) * 1e6:.1f} us per term "
          f"({k * m / dt:.0f} terms/s); e0 {pr.e0_ms / max(pr.e0_launches, 1) * 1e3:.1f} us, "
          f"binv {pr.binv_ms / max(pr.binv_launches, 1) * 1e3:.1f} us, comm {pr.comm_ms / max(pr.comm_launches, 1) * 1e3:.1f} us; "
          f"step-1 kernel {'e0_ck' if ctx.layout_info().e0_kernel else 'e0_lpl'} (timed: e0_lpl {ctx.layout_info().tune_lpl_us:.1f}, "
          f"e0_ck {ctx.layout_info().tune_ck_us:.1f} us)")
    ctx.close()


if __name__ == "__main__":
    main()
