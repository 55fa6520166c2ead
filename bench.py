#!/usr/bin/env python3
"""Headline benchmark: power-series iterations/s of the reduced-camera-system inner solve.

    python bench.py --gpus N --steps K --warmup W        (N=1: plain python; N>1: launched by
    python -m torch.distributed.run --nproc-per-node N ... one rank per GPU)

Workload (BASELINE.json): the shape of BAL venice-1778-993923 (1778 cameras, 993 923 landmarks,
5 001 946 observations), seeded synthetic data (no BAL files offline), fp64, alpha = 0.01,
lambda = 1e-4 (first LM iteration), --power-sc-iterations 20 --eta 0 (exactly 20 terms), the
reference's own definition of the figure: solve_reduced_system_time / linear_solver_iterations
(bal_bundle_adjustment.cpp:355-360).  One "step" = one solve_pOSE (linearization_power_varproj.hpp:
191-237): B^-1(-b) followed by 20 x { E0 x, B^-1, accumulate }, everything resident in HBM.
For N > 1 the landmarks are sharded over the ranks (strong scaling: the problem is fixed) and
each term carries one RCCL all-reduce of the 12*n_cams vector.

Timing: the warm-up is --warmup steps topped up to --warm-seconds (0.5 s) of solves, then the block of
EXACTLY --steps steps (barrier + synchronize on both sides, maximum over the ranks) is run --repeats (5) times back to
back; `value` / `ms_per_step` are the MEDIAN block, `value_min` / `value_max` / `block_ms` the spread, and
`graph_us_per_term` the device time of one replayed term loop per term (event pairs on the library's stream).

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     the E0 (SpMV) kernel pair against the HBM roofline in REAL bytes: `traffic` = PMC-measured HBM bytes
               per application (2 FETCH_SIZE + WRITE_SIZE, profiles/traffic.json) while that file's stamp matches
               the kernel sources, else null; `achieved` = traffic / the pair's HIP-event duration (or, without a
               valid PMC figure, the bytes the two kernels must stream by design, povar_e0_model_bytes: every
               array once -- `basis` says which, `model_GBps` is always given); `frac` = achieved / 8 TB/s (<= 1).
               The SURVEY.md 8(d) stored-tile figure (484 n_obs + 76 n_lms + 288 n_cams per application,
               which the implicit kernels never move) is reported separately as `effective_GBps`.
  cpu_baseline the CPU restatement of the reference algorithm (oracle/, per-camera mutex scatter) on the
               FULL workload: 1 thread and the fastest thread count, CPU model stated

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its N ranks itself
(child `python -m torch.distributed.run`), relays rank 0's JSON line and exits with the children's code.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from povar_amd import capi, synth  # noqa: E402  (loads no GPU state yet)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_e0(n_cams, n_lms, n_obs):
    """SURVEY.md 8(d), stored-tile model, E0 application only (the 144*8*n_cams B^-1 read belongs
    to the B^-1 kernel)."""
    return 484 * n_obs + 76 * n_lms + 288 * n_cams


def algorithmic_bytes_term(n_cams, n_lms, n_obs):
    return 484 * n_obs + 76 * n_lms + 1440 * n_cams


def kernel_source_sha():
    """Stamp of the device code the PMC traffic figures in profiles/traffic.json were measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("povar_kernels.hpp", "povar_kernels_joint.hpp", "povar_kernels_ck.hpp", "povar_kernels_ck_det.hpp", "povar_kernels_ck_joint.hpp",
              "povar_ctx.hpp", "povar_create.hip", "povar_lm.hip", "povar_series.hip", "povar_comm.hip",
              "lpl_layout.hpp", "ck_layout.hpp", "povar_kernels_res.hpp", "res_layout.hpp"):
        with open(os.path.join(ROOT, "povar_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cpus():
    """CPUs this process may use: hardware threads cut by the cgroup CPU quota (the GPU boxes show 256 hardware
    threads under a 16-CPU quota: more than 16 busy threads are throttled, not faster)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                period = int(fh.read())
            if q > 0 and period > 0:
                n = min(n, max(1, -(-q // period)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(prob, alpha, lam, m, max_obs=8_000_000):
    """Oracle (reference-faithful [4k x 16] storage + loop nest of linearization_power_varproj.hpp:364-406,
    per-camera mutex scatter) on the FULL workload: terms/s at 1 thread and at the fastest thread count.
    Workloads above max_obs observations (final-13682: 15 GB of tiles) use the leading landmarks up to
    max_obs and scale by the observation ratio -- said in `sample`."""
    from oracle import povar_oracle as O

    build_note = O.use_native_build()  # the reference's flags, for this host (the portable build is -O3 / x86-64-v3)
    n_l, n_o, scale = prob.n_lms, prob.n_obs, 1.0
    if n_o > max_obs:
        n_l = max(int(np.searchsorted(prob.lm_off, max_obs)), 1)
        n_o = int(prob.lm_off[n_l])
        scale = n_o / prob.n_obs
    orc = O.Oracle(prob.n_cams, prob.lm_off[: n_l + 1], prob.cam_idx[:n_o], prob.obs[:n_o])
    lms = orc.init_landmarks_pose(alpha, prob.cams)
    st, diag2, jls, sigma, ok = orc.stage1_pose(alpha, prob.cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, lam)
    ncpu = effective_cpus()  # "all cores" = what the cgroup lets this process use, not the hardware threads it can see
    # 1 thread: a few terms are enough (the term cost is constant)
    orc.solve_pose(st, hll, binv, b, 1, n_threads=1)
    m1 = max(1, min(m, int(2.0e7 / max(n_o, 1))))
    t0 = time.perf_counter()
    orc.solve_pose(st, hll, binv, b, m1, n_threads=1)
    v1 = m1 / (time.perf_counter() - t0)
    # Thread sweep, ALWAYS reported (threads_tried): 4 / 16 / a quarter of / all usable CPUs, each with one
    # static landmark range per thread and with on-demand chunks (TBB's default auto_partitioner hands out
    # sub-ranges dynamically, linearization_power_varproj.hpp:402-403).  The per-camera mutex (LPV:393-397) makes
    # the scheme contention-bound on hub cameras, so the fastest count is often small; two terms per probe.
    tried = {"1": v1 * scale}
    best = (v1, 1, 0)
    grain_dyn = max(64, n_l // (64 * ncpu))
    # ... and each of them again with per-thread private sums joined after the loop instead of the per-camera mutex
    # ("-private": NOT the reference's scheme for this loop -- its Reductor does that for the column norms,
    # linearization_varproj.hpp:184-209 -- reported so that the mutex's share of the multithreaded slowdown shows)
    for nt in sorted({ncpu, max(ncpu // 4, 1), min(16, ncpu), min(4, ncpu)} - {1}):
        for private in (False, True):
            O.set_e0_scatter(private)
            for grain in (0, grain_dyn):
                O.set_e0_schedule(grain)
                orc.solve_pose(st, hll, binv, b, 1, n_threads=nt)
                t0 = time.perf_counter()
                orc.solve_pose(st, hll, binv, b, 2, n_threads=nt)
                v = 2.0 / (time.perf_counter() - t0)
                tried[f"{nt}" + ("-dynamic" if grain else "") + ("-private" if private else "")] = v * scale
                if v > best[0] and not private:  # the headline baseline stays the reference's scheme
                    best = (v, nt, grain)
    O.set_e0_scatter(False)
    best_private = max([v for k, v in tried.items() if k.endswith("-private")], default=0.0)
    cores, grain = best[1], best[2]
    O.set_e0_schedule(grain)
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.solve_pose(st, hll, binv, b, m, n_threads=cores)
        reps += 1
        if time.perf_counter() - t0 > 8.0 or reps >= 20:
            break
    dt = time.perf_counter() - t0
    O.set_e0_schedule(0)
    vN = reps * m / dt
    cores_note = f", {'on-demand chunks of %d landmarks' % grain if grain else 'static ranges'}" if cores > 1 else ""
    if v1 > vN:  # the mutex scatter often does not scale at all on a hub-heavy graph: the best CPU figure is 1 thread
        cores_note = f" (the {cores}-thread run was slower: {reps * m / dt:.2f} terms/s)" if cores > 1 else ""
        vN, cores = v1, 1
    what = "the full workload" if scale == 1.0 else \
        f"the first {n_l} landmarks / {n_o} observations (all cameras), scaled by {n_o}/{prob.n_obs}"
    return {
        "value": vN * scale,
        "unit": "terms/s",
        "cores": cores,
        "kind": "port",
        "value_1_thread": v1 * scale,
        "threads_tried": {k: round(v, 3) for k, v in tried.items()},
        "value_private_sums": round(best_private, 3),  # best multithreaded rate WITHOUT the per-camera mutex (see threads_tried)
        "build": build_note,
        "value_all_cores": max(tried.get(str(ncpu), 0.0), tried.get(f"{ncpu}-dynamic", 0.0)) if ncpu > 1 else v1 * scale,
        "host_cpus": os.cpu_count() or 1,
        "usable_cpus": ncpu,
        "cpu_model": cpu_model(),
        "sample": f"{what}: {reps} x solve_pOSE of {m} terms with {cores} threads in {dt:.1f} s; "
                  f"{m1} terms with 1 thread at {v1:.2f} terms/s{cores_note}",
    }


def self_launch(argv, n):
    """`python bench.py --gpus N` as the driver calls it: start the N ranks as fresh child processes (this
    process has not touched the GPU), relay rank 0's JSON line, return the children's exit code.  A sharded run
    that dies or stalls with the peer-to-peer term exchange is run once more on the communicator's all-reduce."""
    import signal
    import socket
    import subprocess

    def attempt(extra, limit):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv + extra
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)  # the group this call started, nothing else
            out, _ = p.communicate()
            print(f"[bench] the {n}-rank run did not finish within {limit} s", file=sys.stderr)
        line = None
        for l in (out or "").splitlines():
            if l.startswith("{") and '"metric"' in l:
                line = l
            elif l.strip():
                print(l, file=sys.stderr)
        return p.returncode, line

    rc, line = attempt([], float(os.environ.get("POVAR_BENCH_P2P_LIMIT_S", "1500")))
    if (rc != 0 or line is None) and "--no-p2p" not in argv:
        print("[bench] sharded run failed; once more with --no-p2p (all-reduce per term)", file=sys.stderr)
        rc, line = attempt(["--no-p2p"], None)
    if line is not None:
        print(line)
    return rc if (rc or line is not None) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (one step = one m-term solve; 200 x 1.3 ms: long enough for an external utilisation sampler to see)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed --steps block is run this many times back to back; `value` is the median block")
    ap.add_argument("--warm-seconds", type=float, default=0.5,
                    help="the warm-up is topped up to this much device work whatever --warmup says")
    ap.add_argument("--problem", default="venice-1778", choices=sorted(synth.BAL_SHAPES))
    ap.add_argument("--e0-mode", default="ldsacc", choices=["ldsacc", "implicit", "tiles", "tiles-ldsacc"],
                    help="E0 operator form: implicit tiles + LDS accumulation of hot cameras (default, fastest), "
                         "implicit deterministic, or stored tiles")
    ap.add_argument("--m", type=int, default=20, help="--power-sc-iterations")
    ap.add_argument("--robust-norm", default="NONE", choices=["NONE", "HUBER", "CAUCHY"])
    ap.add_argument("--huber", type=float, default=1.0)
    ap.add_argument("--step", type=int, default=1, choices=[1, 2],
                    help="1: solve_pOSE (headline); 2: solve_joint of the projective refinement (secondary)")
    ap.add_argument("--comm", default="rccl", choices=["rccl", "gloo"],
                    help="exchange steps of the sharded path: RCCL inside the library (default) or a host "
                         "all-reduce over torch.distributed gloo (debug / boxes without a working fabric)")
    ap.add_argument("--popularity", default="zipf1", choices=sorted(synth.POPULARITY),
                    help="camera popularity law of the synthetic graph: zipf1 = the SURVEY 8(d) workload (headline); "
                         "zipf0.5 / uniform = sensitivity variants of the same shape (never the headline)")
    ap.add_argument("--long-track-frac", type=float, default=0.0,
                    help="sensitivity variant: fraction of the observations on landmarks of 65..400 observations")
    ap.add_argument("--no-p2p", action="store_true",
                    help="sharded runs: keep the per-term exchange on the communicator's all-reduce")
    ap.add_argument("--p2p", action="store_true",
                    help="per-term exchange through the peer-to-peer push/reduce kernels (povar_p2p_attach) instead of "
                         "one all-reduce per term; the once-per-solve exchanges keep the communicator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the stored-tile comparison leg")
    ap.add_argument("--with-final", action="store_true",
                    help="also run the final-13682 shape (HUBER 20, 29 M observations: a term loop whose working set is ten "
                         "times the Infinity Cache) in a child process and report its rate and measured fraction under "
                         "`final_13682` (about a minute of problem generation and layout)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))  # nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    # the HIP library first: it owns the GPU data path (HIP + RCCL); torch is only the rendezvous
    capi.lib()
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: E402
        dist.init_process_group(backend="gloo")  # control plane only; data plane is RCCL in the library

    alpha, lam, m = 0.01, 1e-4, args.m
    # the real BAL file when $POVAR_BAL_DIR holds it, else the seeded synthetic shape
    bal_path = synth.find_bal_file(args.problem, os.environ.get("POVAR_BAL_DIR"))
    prob = synth.read_bal_file(bal_path) if bal_path else \
        synth.make_bal_problem(args.problem, args.popularity, args.long_track_frac)
    n_c, n_l, n_o = prob.n_cams, prob.n_lms, prob.n_obs

    n_dev = capi.lib().povar_device_count()
    if n_dev <= 0:
        sys.exit("bench.py needs a HIP device")
    # one rank per GPU; ranks share devices only when there are fewer GPUs than ranks (1-GPU test boxes)
    device = local_rank % n_dev
    if world > n_dev and "POVAR_CU_MASK" not in os.environ:
        # ranks that share a device get disjoint ranges of its CUs (two half devices instead of two processes taking turns
        # on one: a rank that waits inside a kernel for its peer -- the peer-to-peer exchange -- would otherwise keep the peer
        # off the device until its bounded wait runs out)
        share = (world + n_dev - 1) // n_dev
        cus = capi.lib().povar_device_cu_count(device)
        j = local_rank // n_dev
        if cus >= 2 * share:
            os.environ["POVAR_CU_MASK"] = f"{j * (cus // share)}-{(j + 1) * (cus // share) - 1}"
    lb, le = capi.shard_range(prob.lm_off, world, rank)
    ob, oe = int(prob.lm_off[lb]), int(prob.lm_off[le])
    mode = {"implicit": capi.E0_IMPLICIT, "tiles": capi.E0_TILES, "ldsacc": capi.E0_IMPLICIT_LDSACC,
            "tiles-ldsacc": capi.E0_TILES_LDSACC}[args.e0_mode]
    ctx = capi.Context(n_c, prob.lm_off[lb : le + 1] - prob.lm_off[lb], prob.cam_idx[ob:oe],
                       prob.obs[ob:oe], device=device, e0_mode=mode, robust_norm=args.robust_norm,
                       huber=args.huber)
    # steady-state rate: wait for the row placement a host thread of povar_create is still working on (the library
    # starts on the natural row order and swaps at a later linearisation by itself; POVAR_BENCH_NO_WAIT=1 measures that)
    t_wait = time.perf_counter()
    placed = os.environ.get("POVAR_BENCH_NO_WAIT") == "1" or ctx.layout_finalize(wait=True)
    placement_wait_ms = (time.perf_counter() - t_wait) * 1e3
    comm_used = "none"
    if world > 1 or os.environ.get("POVAR_FORCE_COMM"):
        # POVAR_FORCE_COMM=1 exercises the RCCL path with a 1-rank communicator (1-GPU boxes)
        rccl_ok = args.comm == "rccl"
        if rccl_ok:
            try:
                uid = [capi.comm_unique_id() if rank == 0 else None]
                if dist is not None:
                    dist.broadcast_object_list(uid, src=0)
                # RCCL prints a version banner on stdout at communicator creation: keep stdout for the JSON line
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    ctx.comm_init(world, rank, uid[0])
                finally:
                    import ctypes
                    ctypes.CDLL(None).fflush(None)  # the banner sits in C stdio's buffer: flush it to stderr now
                    os.dup2(saved, 1)
                    os.close(saved)
            except capi.PovarError as e:  # pragma: no cover - needs a broken fabric
                print(f"[bench] rank {rank}: RCCL communicator failed: {e}", file=sys.stderr)
                rccl_ok = False
        if dist is not None:
            import torch
            flag = torch.tensor([1 if rccl_ok else 0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            rccl_ok = bool(flag.item())
        if rccl_ok:
            comm_used = "rccl"
        else:
            # exchange steps through torch.distributed(gloo) on host buffers: same sharded algorithm, the
            # collective is no longer RCCL -- reported in config.comm, never silent
            import torch

            def host_allreduce(buf):
                t = torch.from_numpy(buf)
                if dist is not None:
                    dist.all_reduce(t)

            ctx.comm_init_host(world, rank, host_allreduce)
            comm_used = "gloo-host"
            if rank == 0:
                print("[bench] WARNING: exchange steps run over gloo on host buffers, not RCCL", file=sys.stderr)

    term_exchange = "all-reduce" if comm_used != "none" else "none"
    # Per-term exchange of a sharded run: the peer-to-peer push/reduce kernels (no library call inside the captured
    # term loop) unless --no-p2p; they are validated against the communicator's all-reduce below before they are
    # trusted, and every failure (IPC attach, a peer that never delivers, a differing increment) falls back to it.
    # Opt-in (--p2p): its publication order over xGMI has only ever run with both ranks on ONE device, so the
    # driver's multi-GPU run stays on RCCL, the exchange BASELINE.json names.
    want_p2p = comm_used != "none" and not args.no_p2p and args.step == 1 and args.e0_mode == "ldsacc" and args.p2p
    p2p_attached, p2p_note, inc_allreduce = False, "", None
    if want_p2p:
        def all_ok(flag):
            if dist is None:
                return bool(flag)
            import torch
            t = torch.tensor([1 if flag else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())
        handles, mine = [None] * world, None
        try:
            mine = ctx.p2p_export(world)
        except capi.PovarError as e:
            p2p_note = f"export failed: {e}"
        if all_ok(mine is not None):
            if dist is not None:
                dist.all_gather_object(handles, mine)
            else:
                handles = [mine]
            try:
                ctx.p2p_attach(world, rank, handles)
                p2p_attached = True
            except capi.PovarError as e:
                p2p_note = f"attach failed: {e}"
            if not all_ok(p2p_attached):
                if p2p_attached:
                    ctx.p2p_enable(False)
                p2p_attached = False
                p2p_note = p2p_note or "attach failed on another rank"
        else:
            p2p_note = p2p_note or "export failed on another rank"
    ctx.set_cameras(prob.cams)
    ctx.init_landmarks_pose(alpha)
    ok = ctx.linearize_pose(alpha)
    assert ok, "numerical failure during linearization"
    ctx.prepare_pose(lam, capi.POWER_VARPROJ)
    ctx.synchronize()
    # secondary figures: linearize (K3-K6) and prepare_Hb (K7+K8), second call of each
    t_lin = time.perf_counter()
    ok = ctx.linearize_pose(alpha)
    ctx.synchronize()
    t_lin = time.perf_counter() - t_lin
    assert ok
    t_prep = time.perf_counter()
    ctx.prepare_pose(lam, capi.POWER_VARPROJ)
    ctx.synchronize()
    t_prep = time.perf_counter() - t_prep
    if args.step == 2:
        # secondary: the step-2 system at the state step 1 starts from (normalised cameras, X = [x; 1])
        ctx.normalize_joint()
        ok = ctx.linearize_homogeneous()
        assert ok
        ctx.prepare_joint(lam)
        ctx.synchronize()

    def barrier():
        ctx.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    if want_p2p:
        if p2p_attached:
            # one solve through each exchange on the same prepared system; every rank runs both whatever happens
            incs, errs = [], []
            # (first a solve through the all-reduce that is not compared: everything a context does once -- timing its
            # two E0 kernels, capturing the term loop -- happens here, and a host barrier in front of each compared
            # solve brings the ranks to the exchange together: the wait of the push/reduce kernels is bounded)
            ctx.p2p_enable(False)
            try:
                ctx.power_series_pose(m, 0.0, -1.0)
            except capi.PovarError as e:
                errs.append(str(e))
            for on in (True, False):
                ctx.p2p_enable(on)
                barrier()
                try:
                    ctx.power_series_pose(m, 0.0, -1.0)
                    incs.append(ctx.get_increment())
                except capi.PovarError as e:
                    incs.append(None)
                    errs.append(str(e))
            rel = float("inf")
            if incs[0] is not None and incs[1] is not None:
                rel = float(np.linalg.norm(incs[0] - incs[1]) / max(np.linalg.norm(incs[1]), 1e-300))
            if os.environ.get("POVAR_BENCH_P2P_FAIL") and rank == world - 1:
                rel = float("inf")  # test hook: one rank reports a mismatch, every rank must fall back
            good = all_ok(np.isfinite(rel) and rel <= 1e-10)
            inc_allreduce = incs[1]
            ctx.p2p_enable(good)
            if good:
                term_exchange = f"p2p push + local reduce (validated against the all-reduce, rel. diff {rel:.1e})"
            else:
                p2p_note = "; ".join(errs) or f"increment differs from the all-reduce path (rel. diff {rel:.1e})"
        if not term_exchange.startswith("p2p"):
            term_exchange = f"all-reduce (peer-to-peer exchange not used: {p2p_note or 'validation failed on another rank'})"
            if rank == 0:
                print(f"[bench] peer-to-peer term exchange not used: {p2p_note}", file=sys.stderr)

    def run_steps(k):
        for _ in range(k):
            ctx.power_series_pose(m, 0.0, -1.0)

    def agree_max(x):
        """the same number on every rank (the ranks run the same count of collectives)"""
        if dist is None:
            return x
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def warm_up():
        """--warmup steps, then as many more as it takes to keep the device busy for WARM_S seconds: a fresh box needs a few
        hundred milliseconds of work before its clocks and caches are where a long run has them (the driver's 5 warm-up
        solves are 7 ms: BENCH_r05 measured 66.6 us per term where every longer run of the same box class gives 59-61)."""
        barrier()
        t0 = time.perf_counter()
        run_steps(max(args.warmup, 1))
        barrier()
        per_step = agree_max((time.perf_counter() - t0) / max(args.warmup, 1))
        extra = int(min(max(args.warm_seconds / max(per_step, 1e-6) - args.warmup, 0), 20000))
        for at in range(0, extra, 64):  # (at most 64 replayed term loops queued at a time)
            run_steps(min(64, extra - at))
            ctx.synchronize()
        barrier()
        return max(args.warmup, 1) + extra

    def timed_block():
        """EXACTLY --steps steps between two barrier + synchronize pairs; the maximum over the ranks"""
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        return agree_max(time.perf_counter() - t0)

    def timed():
        """the --steps block --repeats times (>= 5 by default), back to back after one warm-up: `value` is the MEDIAN block,
        the fastest and the slowest are reported beside it"""
        n_warm = warm_up()
        return n_warm, [timed_block() for _ in range(max(args.repeats, 1))]

    warm_steps, blocks = timed()
    if term_exchange.startswith("p2p"):
        # the solve is deterministic: the increment the timed loop ended with must still be the validated one; a
        # stale slab anywhere sends every rank back to the all-reduce and the timing is taken again
        inc_now = ctx.get_increment()
        rel2 = float(np.linalg.norm(inc_now - inc_allreduce) / max(np.linalg.norm(inc_allreduce), 1e-300))
        if not all_ok(np.isfinite(rel2) and rel2 <= 1e-10):
            ctx.p2p_enable(False)
            term_exchange = f"all-reduce (peer-to-peer exchange dropped after the timed loop: increment off by {rel2:.1e})"
            if rank == 0:
                print(f"[bench] {term_exchange}", file=sys.stderr)
            warm_steps, blocks = timed()
    dt = float(np.median(blocks))
    # the same --steps block once more with an event pair on the library's stream around every solve (povar_timings, kind
    # "solve"): device time of the whole replayed term loop per term -- beside the host clock of the blocks above it says
    # whether a slow line is a slow device or a slow host
    ctx.timings_enable(True)
    run_steps(args.steps)
    barrier()
    tim = ctx.timings()
    ctx.timings_enable(False)
    graph_us_per_term = 1e3 * tim.solve_ms / max(tim.solve_calls * m, 1)
    # per-kernel durations: the same K steps again with HIP events recorded on the library's stream
    # around every E0 / B^-1 / all-reduce launch (event mode launches kernel by kernel instead of
    # replaying the captured hipGraph, so it is kept out of the headline timing)
    ctx.profile_enable(True)
    run_steps(args.steps)
    barrier()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    inc_main = ctx.get_increment()
    if rank == 0 and os.environ.get("POVAR_BENCH_DUMP_INC"):  # tests: sharded == unsharded increment
        np.save(os.environ["POVAR_BENCH_DUMP_INC"], inc_main)

    terms = args.steps * m
    value = terms / dt
    e0_ms = prof.e0_ms / max(prof.e0_launches, 1)
    binv_ms = prof.binv_ms / max(prof.binv_launches, 1)
    comm_ms = prof.comm_ms / max(prof.comm_launches, 1)
    # the RESIDENT power series (series_res: one launch per solve) has no per-term kernels to put events around -- the event
    # mode above ran the per-term kernels --: a term of the timed loop is then the whole-launch time / m
    series_resident = bool(ctx.layout_info().res_active) and args.step == 1
    per_term_kernels_e0_ms = e0_ms
    if series_resident:
        e0_ms, binv_ms = dt / terms * 1e3, 0.0

    # ONE E0 application on THIS rank (its landmark shard): the bytes its two kernels stream by design, and the
    # SURVEY 8(d) stored-tile figure the implicit kernels are only "effectively" delivering
    loc_l, loc_o = le - lb, oe - ob
    model_lm, model_cm = ctx.e0_model_bytes()
    model_bytes = model_lm + model_cm
    bytes_e0 = algorithmic_bytes_e0(n_c, loc_l, loc_o)
    achieved = model_bytes / (e0_ms * 1e-3) / 1e9 if e0_ms > 0 else 0.0
    effective = bytes_e0 / (e0_ms * 1e-3) / 1e9 if e0_ms > 0 else 0.0
    li0 = ctx.layout_info()
    # "every array once": the chunk kernels read their 18-byte rows on every walk (e0_ck: 2, e0_ck_det: 3)
    once_bytes = model_bytes - (li0.ck_rows * 64 * (10 if li0.ck_packed else 18) * (2 if li0.e0_kernel == 7 else 1)
                                if args.step == 1 and li0.e0_kernel > 0 else 0)  # (step 2's rows are 2-4 bytes: no correction)
    if series_resident:
        # what ONE term of the resident kernel moves: every (workgroup, camera) pair reads the camera's z (192 B of granule
        # pairs), writes its partial record (192 B) and the owner reads it (192 B); every camera's z is published once
        model_lm, model_cm = li0.res_records * 192 * 2, li0.res_records * 192 + n_c * 192
        model_bytes = once_bytes = model_lm + model_cm
        achieved = model_bytes / (e0_ms * 1e-3) / 1e9 if e0_ms > 0 else 0.0
    traffic, traffic_note = None, "no PMC figure for this workload/mode"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as fh:
                tj = json.load(fh)
            # the figure belongs to ONE kernel pair on ONE graph: every knob that changes the kernels or the bytes is in
            # the key (defaults -- step 1, NONE, zipf1, synthetic -- add nothing)
            key = f"{args.problem}:{args.e0_mode}:{world}" + (":step2" if args.step == 2 else "") + \
                (f":{args.robust_norm}" if args.robust_norm != "NONE" else "") + \
                (f":{args.popularity}" if args.popularity != "zipf1" else "") + \
                (f":ltf{args.long_track_frac:g}" if args.long_track_frac > 0 else "") + (":file" if bal_path else "") + \
                (f":ck{ctx.layout_info().e0_kernel}" if args.step == 1 and ctx.layout_info().e0_kernel > 0 else "") + \
                (f":ckh{ctx.layout_info().e0_kernel_h}" if args.step == 2 and ctx.layout_info().e0_kernel_h > 0 else "") + (":resident" if series_resident else "") + \
                (":gather" if os.environ.get("POVAR_DETERMINISTIC") == "1" and (ctx.layout_info().e0_kernel_h if args.step == 2 else ctx.layout_info().e0_kernel) == 0 else "")
            if key in tj:
                if tj.get("_source_sha", {}).get(key) == kernel_source_sha():
                    traffic, traffic_note = tj[key], "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/"
                else:
                    traffic_note = "profiles/traffic.json was measured on other kernel sources (stale): not reported"
        except Exception:
            traffic = None

    out = {
        "metric": "power-series iterations/s (solve_pOSE terms per second)",
        "value": value,
        "unit": "terms/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        # `value` = the median of `repeats` blocks of exactly --steps steps (each between barrier + synchronize pairs, the
        # maximum over the ranks), after `warmup_steps_run` untimed steps (--warmup, topped up to --warm-seconds of work)
        "repeats": len(blocks),
        "value_min": terms / max(blocks),
        "value_max": terms / min(blocks),
        "block_ms": [round(b * 1e3, 4) for b in blocks],
        "warmup_steps_run": warm_steps,
        "graph_us_per_term": graph_us_per_term,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": f"BAL file {os.path.basename(bal_path)}" if bal_path else "synthetic",
        "config": {
            "workload": f"BAL {args.problem} shape ({n_c} cams / {n_l} landmarks / {n_o} obs), "
                        f"{'real BAL file' if bal_path else 'seeded synthetic'}, "
                        f"solve_pOSE with power_sc_iterations={m}, eta=0, lambda={lam}, alpha={alpha}",
            "e0_mode": args.e0_mode,
            "parallelism": f"landmark shards x{world}, one all-reduce (12*n_cams f64) per term" if world > 1
                           else "single GPU",
            "comm": comm_used,
            "term_exchange": term_exchange,
        },
        "spmv_effective_GBps": algorithmic_bytes_term(n_c, n_l, n_o) * value / 1e9,
        "kernel_ms": {"e0": e0_ms, "binv_axpy": binv_ms, "allreduce": comm_ms,
                      "e0_launches": int(prof.e0_launches)},
        "prepare_Hb_ms": t_prep * 1e3,
        "prepare_Hb_effective_GBps": (516 * (oe - ob) + 76 * (le - lb) + 2400 * n_c) / t_prep / 1e9,
        "linearize_ms": t_lin * 1e3,
        "linearize_effective_GBps": 512 * (oe - ob) / t_lin / 1e9,  # the reference's streaming write of every 4x16 tile
        "device_bytes": ctx.device_bytes(),
        "roofline": {
            "bound": "hbm",
            "kernel": "one term of series_res (resident power series: E0 x, B^-1, AXPY in ONE launch per solve)" if series_resident else
                      "E0 x (e0_ck_h_det + cam_cold_sum_binv_h: POVAR_DETERMINISTIC=1)" if args.step == 2 and ctx.layout_info().e0_kernel_h == 2 else
                      "E0 x (e0_ck_det + cam_cold_sum_binv: POVAR_DETERMINISTIC=1)" if args.step == 1 and ctx.layout_info().e0_kernel == 7 else
                      "E0 x (e0_lm_cached<false> + cm_scatter: POVAR_DETERMINISTIC=1, gather form)" if os.environ.get("POVAR_DETERMINISTIC") == "1" else
                      {capi.E0_IMPLICIT: "E0 x (e0_lm_cached<false> + cm_scatter)",
                       capi.E0_IMPLICIT_LDSACC: (("one term of series_res (resident power series: E0 x, B^-1, AXPY in ONE launch per solve)" if series_resident
                                                  else "E0 x (e0_ck_h + cam_cold_sum[_binv]_h)" if args.step == 2 and ctx.layout_info().e0_kernel_h > 0
                                                  else "E0 x (e0_lpl_h + cam_cold_sum[_binv]_h)" if args.step == 2
                                                  else "E0 x (e0_ck + cam_cold_sum[_binv])" if ctx.layout_info().e0_kernel > 0
                                                  else "E0 x (e0_lpl + cam_cold_sum[_binv])")
                                                 if ctx.layout_info().lane_per_landmark else
                                                 "E0 x (e0_lm_cached<true>[_h] + cam_cold_sum[_binv])"),
                       capi.E0_TILES: "E0 x (lm_regular<OpE0Tiles> + cm_scatter)",
                       capi.E0_TILES_LDSACC: "E0 x (e0_tiles_cached + cam_cold_sum[_binv])"}[mode],
            # achieved = HBM bytes of the kernel pair per launch / HIP-event time of the pair.  The bytes are the
            # PMC-measured ones (2 FETCH_SIZE + WRITE_SIZE, profiles/traffic.json) while that file was taken on
            # these kernel sources, otherwise the bytes the pair streams by design (every array once) -- "basis"
            # says which; both rates are always given (traffic_GBps / model_GBps)
            "achieved": (traffic if traffic else model_bytes) / (e0_ms * 1e-3) / 1e9 if e0_ms > 0 else 0.0,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": ((traffic if traffic else model_bytes) / (e0_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if e0_ms > 0 else 0.0,
            "basis": ("measured bytes at the L2's memory side (rocprofv3 PMC passes on these kernel sources; 2 FETCH_SIZE + "
                      "WRITE_SIZE, calibrated: profiles/r04_fetch_calibration.txt)" if traffic
                      else "model bytes of the kernel pair (no PMC figure for these kernel sources / this workload)") +
                     ("; the counters count Infinity-Cache hits and this workload's per-term working set is about the size of "
                      "the 256 MiB cache: a fraction of the HBM peak in fabric bytes, not all of it from HBM"
                      if model_bytes < 600e6 else ""),
            "traffic": traffic,
            "traffic_note": traffic_note,
            "model_bytes_per_launch": model_bytes,
            "model_bytes_lm_kernel": model_lm,
            "model_bytes_cam_kernel": model_cm,
            "model_GBps": achieved,
            "model_frac": achieved / HBM_PEAK_GBPS,
            "traffic_GBps": (traffic / (e0_ms * 1e-3) / 1e9) if traffic else None,
            "traffic_frac": (traffic / (e0_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            # the same yardstick for both step-1 kernels: every array of the operator ONCE (e0_lpl's design bytes; e0_ck
            # reads its row stream on the way forward AND on the way back by design: its model bytes carry the rows twice)
            "once_bytes_per_launch": once_bytes,
            "once_frac": once_bytes / (e0_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if e0_ms > 0 else 0.0,
            # SURVEY 8(d) stored-tile model (the reference's bytes): an EFFECTIVE rate for the implicit kernels
            "algorithmic_bytes_per_launch": bytes_e0,
            "effective_GBps": effective,
            "effective_x_peak": effective / HBM_PEAK_GBPS,
        },
    }
    li = ctx.layout_info()
    out["config"]["e0_layout"] = {"workgroups": li.grid, "lds_camera_slots": li.lds_slots, "global_cameras": li.n_global,
                                  "grid_cameras": li.n_tail, "rows": li.n_rows, "tiles": li.n_tiles,
                                  "lds_resident_obs_frac": 1.0 - li.n_cold / max(li.n_obs, 1),
                                  "assignment": "contiguous landmark ranges, per-workgroup camera sets" if li.strategy else
                                  "rank-based camera grid",
                                  "layout_build_ms": li.create_ms,
                                  "row_placement": {0: "none (natural row order)", 1: "inside povar_create",
                                                    2: "pending on a host thread (kernels on the natural order)",
                                                    3: "on a host thread, swapped in before the timed region"}[li.placement],
                                  "row_placement_ms": li.placement_ms, "row_placement_waited_ms": placement_wait_ms,
                                  # which step-1 E0 kernel ran the timed loop (0: e0_lpl, lane = landmark; > 0: an e0_ck
                                  # instantiation, lane = camera chunk) and how it was chosen
                                  "e0_kernel": li.e0_kernel,
                                  "e0_kernel_choice": {0: "forced", 1: "automatic (not timed yet)",
                                                       2: "automatic: both kernels timed on this problem"}[li.e0_auto],
                                  "e0_tune_us": {"e0_lpl": round(li.tune_lpl_us, 2), "e0_ck": round(li.tune_ck_us, 2)},
                                  # the m-term loop: per-term kernels replayed from a hipGraph, or the resident kernel (one
                                  # launch per solve; contexts of up to 400 k observations) -- chosen by timing both
                                  "series_kernel": "resident (series_res)" if series_resident else "per-term kernels (hipGraph)",
                                  "series_kernel_choice": {0: "forced", 1: "automatic (not timed yet)",
                                                           2: "automatic: both forms timed on this problem"}[li.res_auto],
                                  "series_tune_us_per_term": {"per_term_kernels": round(li.tune_terms_us, 2), "resident": round(li.tune_res_us, 2)},
                                  "resident_series": {"workgroups": li.res_wgs, "wavefronts": li.res_waves, "rows_per_chunk": li.res_rows,
                                                      "chunks_per_lane": li.res_rounds, "partial_records": li.res_records,
                                                      "per_term_kernels_e0_ms": per_term_kernels_e0_ms} if li.res_ready else None,
                                  # step 2: 0: e0_lpl_h, 1: e0_ck_h (its own layout instance: more batches, shorter chunks)
                                  "e0_kernel_step2": li.e0_kernel_h,
                                  "e0_tune_us_step2": {"e0_lpl_h": round(li.tune_lpl_h_us, 2), "e0_ck_h": round(li.tune_ck_h_us, 2)},
                                  "camera_chunks_step2": {"batches": li.ckh_batches, "landmark_slots": li.ckh_slots,
                                                          "chunks": li.ckh_chunks, "own_record_chunks": li.ckh_cold_chunks,
                                                          "lds_stride": li.ckh_stride, "accumulators": li.ckh_accumulators},
                                  "camera_chunks": {"batches": li.ck_batches, "landmark_slots": li.ck_slots, "rows": li.ck_rows,
                                                    # image points as two int32 of micro-units (every observation a six-decimal
                                                    # number, decoded bit for bit) instead of two doubles
                                                    "packed_image_points": bool(li.ck_packed),
                                                    # chunks of cameras without a slot: q per observation in the cold view instead of
                                                    # a partial record per chunk
                                                    "cold_observations_through_the_cold_view": bool(li.ck_cold_q),
                                                    "chunks": li.ck_chunks, "own_record_chunks": li.ck_cold_chunks,
                                                    "partial_records": li.ck_part_rec, "build_ms": round(li.ck_build_ms, 1)}
                                  if li.ck_ready else None,
                                  "term_kernels": "lane per landmark" if li.lane_per_landmark else
                                  "lane per observation (round-1 kernels: under 65 536 observations or POVAR_E0_V1=1)"}
    if not bal_path and (args.popularity != "zipf1" or args.long_track_frac > 0):
        out["config"]["workload"] += f" [SENSITIVITY VARIANT: popularity={args.popularity}, long_track_frac={args.long_track_frac}]"
    if comm_used != "none":  # ncclCommCount of the attached communicator (host-hook runs: its world size)
        out["config"]["rccl_ranks" if comm_used == "rccl" else "host_comm_ranks"] = ctx.comm_ranks()

    if args.step == 2:
        out["metric"] = "power-series iterations/s (solve_joint terms per second, step 2)"
        out["config"]["workload"] = out["config"]["workload"].replace("solve_pOSE", "solve_joint")
    if args.robust_norm != "NONE":
        out["config"]["robust_norm"] = f"{args.robust_norm}({args.huber})"
    if rank == 0 and world == 1 and not args.no_secondary and args.step == 1:
        # secondary leg: the other E0 variant on the same state (also a full-size parity property:
        # both variants must give the same increment)
        tiles_modes = (capi.E0_TILES, capi.E0_TILES_LDSACC)
        other = capi.E0_TILES_LDSACC if mode not in tiles_modes else capi.E0_IMPLICIT_LDSACC
        ctx.set_e0_mode(other)
        run_steps(1)
        ctx.synchronize()
        ctx.profile_enable(True)
        t1 = time.perf_counter()
        run_steps(max(args.steps // 4, 2))
        ctx.synchronize()
        dt2 = time.perf_counter() - t1
        p2 = ctx.profile_get()
        ctx.profile_enable(False)
        inc_other = ctx.get_increment()
        e0_2 = p2.e0_ms / max(p2.e0_launches, 1)
        ach2 = sum(ctx.e0_model_bytes()) / (e0_2 * 1e-3) / 1e9
        out["secondary"] = {
            "e0_mode": "tiles-ldsacc" if other == capi.E0_TILES_LDSACC else "ldsacc",
            "value": max(args.steps // 4, 2) * m / dt2,
            "unit": "terms/s",
            "e0_ms": e0_2,
            "roofline": {"bound": "hbm", "achieved": ach2, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach2 / HBM_PEAK_GBPS},
            "rel_diff_vs_primary": float(np.linalg.norm(inc_other - inc_main) / np.linalg.norm(inc_main)),
        }
        ctx.set_e0_mode(mode)
        # the explicit-Schur-complement solvers on the same linearisation (LinearizorSC: PCG / CHOLESKY);
        # reported beside the headline, never part of `value`
        try:
            ctx.solve_pose_sc(lam, capi.SC_PCG, 0, 500, 1e-6)
            t1 = time.perf_counter()
            _, it_pcg, st_pcg, _ = ctx.solve_pose_sc(lam, capi.SC_PCG, 0, 500, 1e-6)
            dt_pcg = time.perf_counter() - t1
            ctx.solve_pose_sc(lam, capi.SC_CHOLESKY)
            t1 = time.perf_counter()
            x_ch, _, _, rc_ch = ctx.solve_pose_sc(lam, capi.SC_CHOLESKY)
            dt_ch = time.perf_counter() - t1
            n_red = 12 * prob.n_cams
            out["explicit_sc"] = {
                "pcg_iterations_per_s": it_pcg / dt_pcg, "pcg_iterations": it_pcg, "pcg_eta": 1e-6,
                "cholesky_ms": dt_ch * 1e3, "cholesky_dim": n_red,
                "cholesky_tflops": n_red ** 3 / 3 / dt_ch / 1e12, "cholesky_rc": rc_ch,
            }
        except capi.PovarError as e:  # e.g. the dense matrix does not fit: the headline is unaffected
            out["explicit_sc"] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.step == 1:
        out["cpu_baseline"] = cpu_baseline(prob, alpha, lam, m)
    elif rank == 0:
        out["cpu_baseline"] = None

    ctx.close()
    if rank == 0 and world == 1 and not args.no_secondary and args.step == 1 and not bal_path and args.popularity == "zipf1" \
            and args.problem == "venice-1778":
        # The SURVEY 8(d) generator draws cameras with Zipf(1) popularity: ONE hub camera sits in 45 % of the landmarks, which
        # no real BAL problem looks like, and the graph decides which term kernel wins.  A second family with bounded camera
        # degree and locality (cameras on a ring, a landmark seen from neighbouring positions; synth.POPULARITY "local") is
        # therefore reported beside it -- its own process, its own line, never part of `value`.
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--popularity", "local", "--steps", str(max(args.steps // 2, 10)),
                            "--warmup", "2", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True)
        try:
            f = json.loads(r.stdout.strip().splitlines()[-1])
            out.setdefault("secondary", {})["family_local"] = {
                "workload": f["config"]["workload"], "value": f["value"], "unit": f["unit"], "kernel_ms": f["kernel_ms"],
                "roofline": {k: f["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "once_frac", "basis")},
                "e0_kernel": f["config"]["e0_layout"]["e0_kernel"], "e0_tune_us": f["config"]["e0_layout"]["e0_tune_us"],
                "lds_resident_obs_frac": f["config"]["e0_layout"]["lds_resident_obs_frac"]}
        except Exception as e:  # the headline is unaffected
            out.setdefault("secondary", {})["family_local"] = {"error": f"{type(e).__name__}: {e}", "stderr_tail": r.stderr[-500:]}
    if rank == 0 and world == 1 and args.with_final:
        # the one workload of BASELINE.json that is HBM-resident for real (2-3.7 GB per term): its own process, its own line
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--problem", "final-13682", "--robust-norm", "HUBER",
                            "--huber", "20", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"],
                           capture_output=True, text=True)
        try:
            f = json.loads(r.stdout.strip().splitlines()[-1])
            out["final_13682"] = {"value": f["value"], "unit": f["unit"], "ms_per_step": f["ms_per_step"],
                                  "kernel_ms": f["kernel_ms"], "roofline": f["roofline"], "workload": f["config"]["workload"],
                                  "e0_layout": f["config"].get("e0_layout")}
        except Exception as e:  # the headline is unaffected
            out["final_13682"] = {"error": f"{type(e).__name__}: {e}", "stderr_tail": r.stderr[-500:]}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
