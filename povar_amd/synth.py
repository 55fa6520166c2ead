"""Seeded synthetic BAL-shaped problems and the PoVar ``data_custom`` text format.

No BAL files ship with this repo (the reference downloads them from
grail.cs.washington.edu, scripts/download-bal-problems.sh:200), so benchmarks and tests
use generated problems with the exact BAL shapes (scripts/num_ops/bal_numbers.csv:1-5).

The generator follows SURVEY.md section 8(d): landmark degrees ``2 + Geometric`` clipped to
``[2, min(n_cams, 200)]`` and adjusted to hit ``n_obs`` exactly, cameras per landmark drawn
without replacement with Zipf(1.0) popularity over a random camera permutation (hub
cameras), observations = ground-truth pinhole projection + N(0, 0.5^2) px, initial
cameras as ``--create-dataset`` writes them (rows 0-1 ~ N(0,1), row 2 = [0 0 0 1],
bal_problem.cpp:398-407) but from a seeded generator, all values rounded to the 6 decimals
the ``%lf`` writer keeps (bal_problem.cpp:373-434).
"""
from __future__ import annotations

import dataclasses

import numpy as np

# cams / landmarks / observations (scripts/num_ops/bal_numbers.csv; ladybug-49 obs count from
# the BAL site)
BAL_SHAPES = {
    "ladybug-49": (49, 7776, 31843),
    "trafalgar-257": (257, 65132, 225911),
    "venice-1778": (1778, 993923, 5001946),
    "final-13682": (13682, 4456117, 28987644),
}
BAL_SEEDS = {"ladybug-49": 49, "trafalgar-257": 257, "venice-1778": 1778, "final-13682": 13682}


@dataclasses.dataclass
class Problem:
    """Loaded problem in the layout of include/povar_hip.h (v already negated)."""

    n_cams: int
    n_lms: int
    lm_off: np.ndarray  # int32 [n_lms + 1]
    cam_idx: np.ndarray  # int32 [n_obs], ascending inside a landmark
    obs: np.ndarray  # float64 [n_obs, 2]
    cams: np.ndarray  # float64 [n_cams, 12] initial space matrices
    lms: np.ndarray  # float64 [n_lms, 3] (arbitrary; overwritten by the VarPro init)

    @property
    def n_obs(self) -> int:
        return int(self.cam_idx.shape[0])


def _degrees(rng, n_cams, n_lms, n_obs):
    kmax = min(n_cams, 200)
    mean = n_obs / n_lms
    assert 2 <= mean <= kmax
    p = 1.0 / max(mean - 1.0, 1e-9)
    k = 1 + rng.geometric(min(p, 1.0), size=n_lms).astype(np.int64)
    k = np.clip(k, 2, kmax)
    # adjust to hit n_obs exactly
    diff = int(n_obs - k.sum())
    while diff != 0:
        step = 1 if diff > 0 else -1
        ok = np.flatnonzero((k < kmax) if step > 0 else (k > 2))
        take = rng.choice(ok, size=min(abs(diff), ok.size), replace=False)
        k[take] += step
        diff = int(n_obs - k.sum())
    return k


POPULARITY = {"zipf1": 1.0, "zipf0.5": 0.5, "uniform": 0.0, "local": None}


def _sample_cameras_local(rng, n_cams, lm_off, hub_frac=0.1):
    """A graph with the locality of a real reconstruction (the "local" sensitivity variant): the cameras sit on a ring
    (a trajectory), landmark l is seen from around position l / n_lms of it -- so neighbouring landmarks of the file
    share cameras, as they do in the BAL files (bal_problem.cpp:183-303 keeps the file's landmark order) -- by cameras
    at two-sided geometric (Laplace) offsets of scale 24 positions -- how many cameras see one region of a scene does
    not grow with the size of the collection --, and `hub_frac` of the observations go to Zipf(1) hub cameras anywhere
    on the ring."""
    n_obs = int(lm_off[-1])
    n_lms = lm_off.shape[0] - 1
    lm_of = np.repeat(np.arange(n_lms, dtype=np.int64), np.diff(lm_off))
    centre = (lm_of * n_cams) // n_lms
    scale = min(24.0, max(2.0, n_cams / 8.0))
    w = 1.0 / np.arange(1, n_cams + 1)
    cdf = np.cumsum(w / w.sum())
    cdf[-1] = 1.0
    perm = rng.permutation(n_cams)

    def draw(idx):
        off = np.rint(rng.laplace(scale=scale, size=idx.shape[0])).astype(np.int64)
        c = (centre[idx] + off) % n_cams
        hub = rng.random(idx.shape[0]) < hub_frac
        c[hub] = perm[np.searchsorted(cdf, rng.random(int(hub.sum())), side="right")]
        return c

    cam = draw(np.arange(n_obs))
    active = np.arange(n_obs)
    while True:
        key = lm_of[active] * n_cams + cam[active]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        dup = np.zeros(active.shape[0], dtype=bool)
        dup[order[1:]] = ks[1:] == ks[:-1]
        if not dup.any():
            break
        bad_lm = np.unique(lm_of[active[dup]])
        redraw = active[dup]
        cam[redraw] = draw(redraw)
        mask = np.zeros(n_lms, dtype=bool)
        mask[bad_lm] = True
        active = active[mask[lm_of[active]]]
    order = np.lexsort((cam, lm_of))
    return cam[order].astype(np.int32)


def _sample_cameras(rng, n_cams, lm_off, zipf_s=1.0):
    """Zipf(s)-weighted sampling without replacement per landmark (redraw duplicates); s = 1 is the SURVEY 8(d)
    workload, s = 0 a uniform popularity (no hub cameras at all)."""
    n_obs = int(lm_off[-1])
    n_lms = lm_off.shape[0] - 1
    perm = rng.permutation(n_cams)
    w = 1.0 / np.arange(1, n_cams + 1) ** zipf_s
    cdf = np.cumsum(w / w.sum())
    cdf[-1] = 1.0
    lm_of = np.repeat(np.arange(n_lms, dtype=np.int64), np.diff(lm_off))
    cam = perm[np.searchsorted(cdf, rng.random(n_obs), side="right")].astype(np.int64)
    active = np.arange(n_obs)
    while True:
        key = lm_of[active] * n_cams + cam[active]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        dup = np.zeros(active.shape[0], dtype=bool)
        dup[order[1:]] = ks[1:] == ks[:-1]
        if not dup.any():
            break
        # keep working only on landmarks that still have duplicates
        bad_lm = np.unique(lm_of[active[dup]])
        redraw = active[dup]
        cam[redraw] = perm[np.searchsorted(cdf, rng.random(redraw.shape[0]), side="right")]
        mask = np.zeros(n_lms, dtype=bool)
        mask[bad_lm] = True
        active = active[mask[lm_of[active]]]
    # ascending camera index inside each landmark (std::map order, bal_problem.hpp:226)
    order = np.lexsort((cam, lm_of))
    return cam[order].astype(np.int32)


def make_problem(n_cams, n_lms, n_obs, seed=0, noise_px=0.5, popularity="zipf1", long_track_frac=0.0,
                 init="random", init_noise=0.0) -> Problem:
    """popularity: camera popularity law of the graph ("zipf1" = SURVEY 8(d), "zipf0.5", "uniform"; "local" = cameras
    on a ring, landmarks seen by neighbouring cameras, 10 % hub observations: the locality of a real reconstruction).
    init: "random" = the reference's --create-dataset cameras (rows 0-1 ~ N(0,1), row 2 = [0 0 0 1]); "gt" = the
    ground-truth projection matrices K[R|t], each divided by its mean depth (the pOSE affine term wants P_2 X ~ 1)
    and perturbed entrywise by a relative N(0, init_noise^2) -- a start inside the basin of both steps, used by the
    known-answer tests at BASELINE sizes (a converged run must end on the chi-square floor of `noise_px`).
    long_track_frac: fraction of the observations moved onto landmarks of 65..min(n_cams, 400) observations
    (real photo collections have such tracks; the SURVEY 8(d) degree law caps them at 49 for venice)."""
    rng = np.random.default_rng(seed)
    k = _degrees(rng, n_cams, n_lms, n_obs)
    if long_track_frac > 0:
        # lengthen the first landmarks to 65..kmax observations and take the same number of observations away
        # from the others (never below 2), keeping n_lms and n_obs
        kmax = min(n_cams, 400)
        want = int(long_track_frac * n_obs)
        n_long = max(1, want // ((65 + kmax) // 2))
        new_k = rng.integers(65, kmax + 1, size=n_long)
        delta = int(new_k.sum() - k[:n_long].sum())
        k[:n_long] = new_k
        i = n_long
        while delta > 0:
            take = np.flatnonzero(k[n_long:] > 2) + n_long
            take = rng.choice(take, size=min(delta, take.size), replace=False)
            k[take] -= 1
            delta -= take.size
        rng.shuffle(k)
    lm_off = np.zeros(n_lms + 1, dtype=np.int64)
    np.cumsum(k, out=lm_off[1:])
    cam_idx = _sample_cameras_local(rng, n_cams, lm_off) if popularity == "local" else \
        _sample_cameras(rng, n_cams, lm_off, POPULARITY[popularity])
    lm_of = np.repeat(np.arange(n_lms), k)

    # ground truth: points in a unit cube, cameras 5-15 units away looking at it
    X = rng.random((n_lms, 3)) - 0.5
    f = rng.uniform(500.0, 2000.0, size=n_cams)
    d = rng.normal(size=(n_cams, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    centre = -d * rng.uniform(5.0, 15.0, size=(n_cams, 1))
    up = rng.normal(size=(n_cams, 3))
    xax = np.cross(up, d)
    xax /= np.linalg.norm(xax, axis=1, keepdims=True)
    yax = np.cross(d, xax)
    R = np.stack([xax, yax, d], axis=1)  # rows: camera axes, z looks at the cube
    t = -np.einsum("cij,cj->ci", R, centre)
    pc = np.einsum("oij,oj->oi", R[cam_idx], X[lm_of]) + t[cam_idx]
    uv = f[cam_idx, None] * pc[:, :2] / pc[:, 2:3]
    uv += rng.normal(scale=noise_px, size=uv.shape)
    uv = np.round(uv, 6)  # what "%lf" keeps (bal_problem.cpp:373-375)

    cams = np.zeros((n_cams, 12))
    cams[:, :8] = rng.normal(size=(n_cams, 8))
    cams[:, 11] = 1.0
    cams = np.round(cams, 6)
    lms = np.round(rng.normal(size=(n_lms, 3)), 6)
    if init == "gt":
        K = np.zeros((n_cams, 3, 3))
        K[:, 0, 0] = K[:, 1, 1] = f
        K[:, 2, 2] = 1.0
        P = np.einsum("cij,cjk->cik", K, np.concatenate([R, t[:, :, None]], axis=2))
        depth = np.bincount(cam_idx, weights=pc[:, 2], minlength=n_cams) / np.maximum(np.bincount(cam_idx, minlength=n_cams), 1)
        P /= np.where(depth > 0, depth, 1.0)[:, None, None]
        P *= 1.0 + init_noise * np.random.default_rng(seed + 77).normal(size=P.shape)
        cams = np.round(P.reshape(n_cams, 12), 6)
    elif init != "random":
        raise ValueError(init)
    return Problem(n_cams, n_lms, lm_off.astype(np.int32), cam_idx, np.ascontiguousarray(uv), cams, lms)


def make_bal_problem(name: str, popularity="zipf1", long_track_frac=0.0, **kw) -> Problem:
    """Seeded synthetic problem with the exact shape of a BAL problem (BASELINE.json configs).  The defaults are the
    SURVEY 8(d) workload; the other popularity laws / a long-track tail are sensitivity variants of the same shape."""
    n_c, n_l, n_o = BAL_SHAPES[name]
    return make_problem(n_c, n_l, n_o, seed=BAL_SEEDS[name], popularity=popularity, long_track_frac=long_track_frac, **kw)


def write_data_custom(path: str, prob: Problem) -> None:
    """PoVar ``data_custom/<name>`` text format (written bal_problem.cpp:334-434, read by
    ``load_bal_eccv`` bal_problem.cpp:193-268): header ``n_c n_l n_o``; ``cam lm x y`` per
    observation in the ORIGINAL BAL sign (the loader negates y, bal_problem.cpp:240); 15
    values per camera (row-major 3x4 + f k1 k2); 3 values per landmark."""
    lm_of = np.repeat(np.arange(prob.n_lms), np.diff(prob.lm_off))
    with open(path, "w") as fh:
        fh.write(f"{prob.n_cams} {prob.n_lms} {prob.n_obs}")
        for c, l, (u, v) in zip(prob.cam_idx, lm_of, prob.obs):
            fh.write(f"\n{c} {l} {u:.6f} {-v:.6f}")
        for P in prob.cams:
            for x in P:
                fh.write(f"\n{x:.6f}")
            fh.write("\n1.000000\n0.000000\n0.000000")
        for p in prob.lms:
            for x in p:
                fh.write(f"\n{x:.6f}")
        fh.write("\n")


def read_data_custom(path: str) -> Problem:
    """Python mirror of ``BalProblem::load_bal_eccv`` (bal_problem.cpp:183-303): parses the
    file, negates y, groups observations per landmark with ascending camera index and rejects
    duplicate (camera, landmark) pairs (bal_problem.cpp:227)."""
    with open(path) as fh:
        tok = fh.read().split()
    n_c, n_l, n_o = int(tok[0]), int(tok[1]), int(tok[2])
    if min(n_c, n_l, n_o) <= 0:
        raise ValueError("invalid header")
    body = np.array(tok[3 : 3 + 4 * n_o], dtype=np.float64).reshape(n_o, 4)
    cam = body[:, 0].astype(np.int64)
    lm = body[:, 1].astype(np.int64)
    if cam.min() < 0 or cam.max() >= n_c or lm.min() < 0 or lm.max() >= n_l:
        raise ValueError("index out of range")
    uv = body[:, 2:4].copy()
    uv[:, 1] = -uv[:, 1]
    order = np.lexsort((cam, lm))
    cam, lm, uv = cam[order], lm[order], uv[order]
    if np.any((cam[1:] == cam[:-1]) & (lm[1:] == lm[:-1])):
        raise ValueError("duplicate observation")
    lm_off = np.zeros(n_l + 1, dtype=np.int64)
    np.cumsum(np.bincount(lm, minlength=n_l), out=lm_off[1:])
    p = 3 + 4 * n_o
    cp = np.array(tok[p : p + 15 * n_c], dtype=np.float64).reshape(n_c, 15)
    p += 15 * n_c
    lms = np.array(tok[p : p + 3 * n_l], dtype=np.float64).reshape(n_l, 3)
    return Problem(n_c, n_l, lm_off.astype(np.int32), cam.astype(np.int32),
                   np.ascontiguousarray(uv), np.ascontiguousarray(cp[:, :12]), lms)


BAL_FILES = {  # scripts/download-bal-problems.sh:30-120 (file names on grail.cs.washington.edu)
    "ladybug-49": "problem-49-7776-pre.txt",
    "trafalgar-257": "problem-257-65132-pre.txt",
    "venice-1778": "problem-1778-993923-pre.txt",
    "final-13682": "problem-13682-4456117-pre.txt",
}


def read_bal_file(path: str, seed: int = 38401) -> Problem:
    """A real BAL problem, either in the ORIGINAL format (9 parameters per camera) -- the initial cameras
    are then drawn as ``--create-dataset`` does (rows 0-1 ~ N(0,1), row 2 = [0 0 0 1],
    bal_problem.cpp:398-407) from a generator seeded with ``seed`` -- or in PoVar's ``data_custom`` format
    (15 values per camera).  ``.bz2`` files are read directly.  Observations are negated in y, grouped per
    landmark with ascending camera index, duplicates rejected, as ``load_bal_eccv`` does."""
    import bz2
    opener = bz2.open if path.endswith(".bz2") else open
    with opener(path, "rt") as fh:
        data = np.array(fh.read().split(), dtype=np.float64)
    n_c, n_l, n_o = int(data[0]), int(data[1]), int(data[2])
    if min(n_c, n_l, n_o) <= 0:
        raise ValueError("invalid header")
    body = data[3 : 3 + 4 * n_o].reshape(n_o, 4)
    rest = data[3 + 4 * n_o :]
    cam = body[:, 0].astype(np.int64)
    lm = body[:, 1].astype(np.int64)
    if cam.min() < 0 or cam.max() >= n_c or lm.min() < 0 or lm.max() >= n_l:
        raise ValueError("index out of range")
    uv = body[:, 2:4].copy()
    uv[:, 1] = -uv[:, 1]
    order = np.lexsort((cam, lm))
    cam, lm, uv = cam[order], lm[order], uv[order]
    if np.any((cam[1:] == cam[:-1]) & (lm[1:] == lm[:-1])):
        raise ValueError("duplicate observation")
    lm_off = np.zeros(n_l + 1, dtype=np.int64)
    np.cumsum(np.bincount(lm, minlength=n_l), out=lm_off[1:])
    if rest.shape[0] == 15 * n_c + 3 * n_l:      # data_custom
        cams = rest[: 15 * n_c].reshape(n_c, 15)[:, :12].copy()
        lms = rest[15 * n_c :].reshape(n_l, 3)
    elif rest.shape[0] == 9 * n_c + 3 * n_l:     # original BAL
        rng = np.random.default_rng(seed)
        cams = np.zeros((n_c, 12))
        cams[:, :8] = rng.normal(size=(n_c, 8))
        cams[:, 11] = 1.0
        lms = rest[9 * n_c :].reshape(n_l, 3)
    else:
        raise ValueError("neither an original BAL file nor a data_custom file")
    return Problem(n_c, n_l, lm_off.astype(np.int32), cam.astype(np.int32), np.ascontiguousarray(uv),
                   np.ascontiguousarray(cams), np.ascontiguousarray(lms))


def find_bal_file(name: str, directory: str | None):
    """``$POVAR_BAL_DIR`` lookup for bench.py: <dir>/[data_custom/]<file>[.bz2]."""
    import os
    if not directory:
        return None
    for sub in ("data_custom", ""):
        for ext in ("", ".bz2"):
            p = os.path.join(directory, sub, BAL_FILES[name] + ext)
            if os.path.exists(p):
                return p
    return None
