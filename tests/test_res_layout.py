"""Host-only invariants of the layout builder of the resident power series (povar_amd/csrc/res_layout.hpp, the layout of
series_res: one launch per solve_pOSE, sc/linearization_power_varproj.hpp:191-237) through tests/cpp/res_layout_check.cpp:
every observation in exactly one lane chunk of its camera, landmark slots that name its landmark inside the workgroup,
accumulator slots consistent with the workgroup's camera set, partial records camera-major and used once, every camera
owned by exactly one workgroup, LDS capacity respected.  Runs without a GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "res_layout_check")


def _run(tmp_path, n_cams, lm_off, cam_idx, obs, wgs, n_waves=8, rounds=2, hmin=1, hmax=4, ls_max=2, order=-1):
    src = [os.path.join(ROOT, "tests", "cpp", "res_layout_check.cpp"), os.path.join(ROOT, "povar_amd", "csrc", "res_layout.hpp"),
           os.path.join(ROOT, "povar_amd", "csrc", "lpl_layout.hpp")]
    if not os.path.exists(BIN) or any(os.path.getmtime(BIN) < os.path.getmtime(s) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-B", "../../build/res_layout_check"],
                              stdout=subprocess.DEVNULL)
    f = [str(tmp_path / n) for n in ("lm_off.bin", "cam_idx.bin", "obs.bin")]
    np.ascontiguousarray(lm_off, dtype=np.int32).tofile(f[0])
    np.ascontiguousarray(cam_idx, dtype=np.int32).tofile(f[1])
    np.ascontiguousarray(obs, dtype=np.float64).tofile(f[2])
    r = subprocess.run([BIN, str(n_cams)] + f + [str(wgs), str(n_waves), str(rounds), str(hmin), str(hmax), str(ls_max), str(order)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


BIG = dict(hmin=4)  # four rows per chunk whatever the problem needs
WIDE = dict(n_waves=16, rounds=1, hmax=2, ls_max=1)  # a 1024-thread shape (not instantiated: 128 VGPRs per lane)


@pytest.mark.parametrize("wgs,kw,must_fit", [(256, {}, True), (256, WIDE, True), (32, WIDE, False), (32, BIG, True), (7, BIG, False), (1, BIG, False),
                                             (90, dict(hmax=1), False), (16, dict(n_waves=8, rounds=1, hmin=8, hmax=8, ls_max=2), False)])
def test_res_layout_invariants_medium(tmp_path, wgs, kw, must_fit):
    from povar_amd import synth
    p = synth.make_problem(300, 20000, 90000, seed=5)
    fits = []
    for order in (-1, 0, 1):
        s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, wgs, order=order, **kw)
        assert s["ok"] == 1
        fits.append(s["fits"])
        if s["fits"]:
            assert s["lds_bytes"] <= 160 * 1024 and s["W"] <= wgs
    if must_fit:
        assert all(fits)


def test_res_layout_long_tracks_and_single_observation_landmarks(tmp_path):
    rng = np.random.default_rng(3)
    n_c = 700
    ks = np.concatenate([[1, 1, 2, 650, 130, 64, 65, 9, 8], rng.integers(1, 12, size=3000)])
    lm_off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    w = 1.0 / np.arange(1, n_c + 1)
    cam_idx = np.concatenate([np.sort(rng.choice(n_c, k, replace=False, p=w / w.sum())) for k in ks]).astype(np.int32)
    obs = rng.normal(size=(cam_idx.shape[0], 2))
    for wgs, kw in ((40, BIG), (64, {}), (64, WIDE)):
        s = _run(tmp_path, n_c, lm_off, cam_idx, obs, wgs, **kw)
        assert s["ok"] == 1 and s["fits"] == 1


@pytest.mark.parametrize("name,wgs", [("ladybug-49", 32), ("trafalgar-257", 221)])
def test_res_layout_baseline_shapes(tmp_path, name, wgs):
    from povar_amd import synth
    p = synth.make_bal_problem(name)
    s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, wgs)
    assert s["ok"] == 1 and s["fits"] == 1 and s["H"] <= 2, s
    print(s)


def test_res_layout_venice_shard_of_eight(tmp_path):
    """rank 0's landmark shard of venice-1778 at world = 8 (BASELINE config 4): fits 256 workgroups of 512 lanes with two
    chunks of four rows each; the full problem does not (and says so instead of building something)."""
    from povar_amd import capi, synth
    p = synth.make_bal_problem("venice-1778")
    lb, le = capi.shard_range(p.lm_off, 8, 0)
    ob, oe = int(p.lm_off[lb]), int(p.lm_off[le])
    s = _run(tmp_path, p.n_cams, p.lm_off[lb:le + 1] - p.lm_off[lb], p.cam_idx[ob:oe], p.obs[ob:oe], 256, **BIG)
    print(s)
    assert s["ok"] == 1 and s["fits"] == 1, s
    full = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 256, **BIG)
    assert full["ok"] == 1 and full["fits"] == 0
