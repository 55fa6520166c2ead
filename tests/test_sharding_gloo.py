"""N > 1 path on CPU: two gloo ranks, landmarks sharded with the product's povar_shard_range,
the per-term exchange (one all-reduce of the 12*n_cams vector) with the oracle standing in for
the device kernels.  Sharded result == unsharded result to reduction-order tolerance."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from povar_amd import capi, synth
    from oracle import povar_oracle as O
    p = synth.make_problem(12, 400, 1700, seed=4)
    alpha, lam, m = 0.01, 1e-4, 8
    lb, le = capi.shard_range(p.lm_off, world, rank)
    ob, oe = int(p.lm_off[lb]), int(p.lm_off[le])
    orc = O.Oracle(p.n_cams, p.lm_off[lb:le + 1] - p.lm_off[lb], p.cam_idx[ob:oe], p.obs[ob:oe])
    lms = orc.init_landmarks_pose(alpha, p.cams)          # landmark-local, no exchange
    st, ok = orc.linearize_pose(alpha, p.cams, lms)
    diag2 = torch.from_numpy(orc.jp_diag2_pose(st))       # exchange 1: per linearisation
    dist.all_reduce(diag2)
    jls = orc.scale_jl_cols_pose(st)
    sigma = 1.0 / (1e-5 + np.sqrt(diag2.numpy()))
    orc.scale_jp_cols_pose(st, sigma)
    hll, b_part, hpp_part = orc.prepare_hb_pose(st, 0.0)  # partial sums; B^-1 rebuilt after the exchange
    # exchange 2: per solve -- b and the Hpp blocks.  prepare_hb_pose inverted (Hpp + 0)^-1 per rank,
    # which is not additive, so rebuild Hpp from the tiles instead:
    hpp = np.zeros((p.n_cams, 12, 12))
    for i, c in enumerate(orc.cam_idx):
        J = st[4 * i:4 * i + 4, :12]
        hpp[c] += J.T @ J
    hpp_t, b_t = torch.from_numpy(hpp), torch.from_numpy(b_part)
    dist.all_reduce(hpp_t)
    dist.all_reduce(b_t)
    binv = np.stack([np.linalg.inv(h + lam * np.eye(12)) for h in hpp_t.numpy()]).reshape(p.n_cams, 144)
    b = b_t.numpy()
    accum = orc.right_mul_b_inv(binv, -b)
    tmp = accum.copy()
    for _ in range(m):
        y = torch.from_numpy(orc.right_mul_e0_pose(st, hll, tmp))  # exchange 3: once per power term
        dist.all_reduce(y)
        tmp = orc.right_mul_b_inv(binv, y.numpy())
        accum += tmp
    if rank == 0:
        np.save(out, accum)
    dist.destroy_process_group()


def test_two_rank_sharded_series_matches_single(tmp_path):
    sys.path.insert(0, ROOT)
    from povar_amd import synth
    from oracle import povar_oracle as O
    O.build()
    out = str(tmp_path / "acc.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    sharded = np.load(out)
    p = synth.make_problem(12, 400, 1700, seed=4)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    lms = orc.init_landmarks_pose(0.01, p.cams)
    st, diag2, jls, sigma, ok = orc.stage1_pose(0.01, p.cams, lms)
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, 1e-4)
    ref, _, _, _ = orc.solve_pose(st, hll, binv, b, 8)
    assert np.linalg.norm(sharded - ref) / np.linalg.norm(ref) < 1e-11


def _worker_sc(rank, world, port, out):
    """Explicit-SC solvers on landmark shards: S and b are additive over the shards once the pose damping is
    taken out (it is added by one rank only, as run_cholesky / cam_build_sc do); PCG then runs replicated."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from povar_amd import capi, synth
    from oracle import povar_oracle as O
    p = synth.make_problem(12, 400, 1700, seed=4)
    alpha, lam = 0.01, 1e-3
    lb, le = capi.shard_range(p.lm_off, world, rank)
    ob, oe = int(p.lm_off[lb]), int(p.lm_off[le])
    orc = O.Oracle(p.n_cams, p.lm_off[lb:le + 1] - p.lm_off[lb], p.cam_idx[ob:oe], p.obs[ob:oe])
    lms = orc.init_landmarks_pose(alpha, p.cams)
    st, ok = orc.linearize_pose(alpha, p.cams, lms)
    diag2 = torch.from_numpy(orc.jp_diag2_pose(st))
    dist.all_reduce(diag2)
    orc.scale_jp_cols_pose(st, 1.0 / (1e-5 + np.sqrt(diag2.numpy())))
    S, b = orc.get_hb_pose(st, lam if rank == 0 else 0.0)   # damping once over the ranks
    S_t, b_t = torch.from_numpy(S), torch.from_numpy(b)
    dist.all_reduce(S_t)
    dist.all_reduce(b_t)
    S, b = S_t.numpy(), b_t.numpy()
    x, it, status = orc.pcg(S, b, orc.block_jacobi_inverse(S, 12), eta=1e-2)
    xc, bad = orc.cholesky_solve(S, b)
    if rank == 0:
        np.savez(out, pcg=x, it=it, status=status, chol=xc, bad=bad)
    dist.destroy_process_group()


def test_two_rank_sharded_explicit_sc_matches_single(tmp_path):
    sys.path.insert(0, ROOT)
    from povar_amd import synth
    from oracle import povar_oracle as O
    O.build()
    out = str(tmp_path / "sc.npz")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_sc, args=(2, port, out), nprocs=2, join=True)
    g = np.load(out)
    p = synth.make_problem(12, 400, 1700, seed=4)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    lms = orc.init_landmarks_pose(0.01, p.cams)
    st, ok = orc.linearize_pose(0.01, p.cams, lms)
    orc.scale_jp_cols_pose(st, 1.0 / (1e-5 + np.sqrt(orc.jp_diag2_pose(st))))
    S, b = orc.get_hb_pose(st, 1e-3)
    x, it, status = orc.pcg(S, b, orc.block_jacobi_inverse(S, 12), eta=1e-2)
    xc, bad = orc.cholesky_solve(S, b)
    assert (int(g["it"]), int(g["status"]), int(g["bad"])) == (it, status, bad)
    assert np.linalg.norm(g["pcg"] - x) / np.linalg.norm(x) < 1e-9
    assert np.linalg.norm(g["chol"] - xc) / np.linalg.norm(xc) < 1e-9
