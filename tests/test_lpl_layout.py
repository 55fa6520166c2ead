"""Host-only invariants of the lane-per-landmark layout builder (povar_amd/csrc/lpl_layout.hpp) through the
checker binary tests/cpp/lpl_layout_check.cpp: every observation placed exactly once with its camera's slot or a
valid cold position, tiles longest first, partial records one per (workgroup, slot), LDS capacity respected,
workgroup loads balanced.  Runs without a GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "lpl_layout_check")


def _run(tmp_path, n_cams, lm_off, cam_idx, obs, grid, n_acc, env=None):
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "../../build/lpl_layout_check"],
                              stdout=subprocess.DEVNULL)
    f = [str(tmp_path / n) for n in ("lm_off.bin", "cam_idx.bin", "obs.bin")]
    np.ascontiguousarray(lm_off, dtype=np.int32).tofile(f[0])
    np.ascontiguousarray(cam_idx, dtype=np.int32).tofile(f[1])
    np.ascontiguousarray(obs, dtype=np.float64).tofile(f[2])
    r = subprocess.run([BIN, str(n_cams)] + f + [str(grid), str(n_acc)], capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("grid,n_acc", [(256, 568), (256, 64), (7, 40), (1, 568), (12, 16)])
def test_layout_invariants_medium(tmp_path, grid, n_acc):
    from povar_amd import synth
    p = synth.make_problem(300, 20000, 90000, seed=5)
    s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, grid, n_acc)
    assert s["ok"] == 1 and s["max_slots"] <= n_acc
    if p.n_cams <= n_acc:
        assert s["cold"] == 0 and s["n_global"] == p.n_cams


def test_layout_long_tracks_and_single_observation_landmarks(tmp_path):
    rng = np.random.default_rng(3)
    n_c = 700
    ks = np.concatenate([[1, 1, 2, 650, 130, 64, 65, 9, 8], rng.integers(1, 12, size=3000)])
    lm_off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    w = 1.0 / np.arange(1, n_c + 1)
    cam_idx = np.concatenate([np.sort(rng.choice(n_c, k, replace=False, p=w / w.sum())) for k in ks]).astype(np.int32)
    obs = rng.normal(size=(cam_idx.shape[0], 2))
    s = _run(tmp_path, n_c, lm_off, cam_idx, obs, 16, 100)
    assert s["ok"] == 1


def test_layout_venice_shape_coverage(tmp_path):
    """The BASELINE shape: the camera grid leaves ~1 % of the observations cold (14 % with the global hot set only)."""
    from povar_amd import synth
    p = synth.make_bal_problem("venice-1778")
    s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 256, 568)
    assert s["ok"] == 1 and s["cold_frac"] < 0.03 and s["pad_frac"] < 0.05
    assert s["wg_rows_max"] <= 1.05 * s["wg_rows_min"] + 8
    g = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 256, 568, env={"POVAR_LPL_NOGRID": "1"})
    assert g["ok"] == 1 and 0.10 < g["cold_frac"] < 0.18


def test_layout_does_not_depend_on_the_thread_count(tmp_path):
    """Every parallel phase of build_lpl cuts its work into independent pieces (landmark chunks, workgroups): the layout
    built on 1, 3 and 8 host threads is the same, byte for byte."""
    from povar_amd import synth
    p = synth.make_problem(300, 20000, 90000, seed=5)
    fps = {t: _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 64, 120, env={"POVAR_LAYOUT_THREADS": str(t)})["fingerprint"]
           for t in (1, 3, 8)}
    assert len(set(fps.values())) == 1, fps


def test_layout_strategy_follows_the_graph(tmp_path):
    """A graph with locality (cameras on a ring, landmarks seen by neighbouring cameras, file order = position) gets
    contiguous landmark ranges with per-workgroup camera sets and is almost entirely LDS-resident; the SURVEY 8(d) Zipf
    graph (no locality) keeps the rank-based camera grid.  Both strategies satisfy every layout invariant on both."""
    from povar_amd import synth
    loc = synth.make_problem(1200, 60000, 300000, seed=11, popularity="local")
    zipf = synth.make_problem(1200, 60000, 300000, seed=11)
    a = _run(tmp_path, loc.n_cams, loc.lm_off, loc.cam_idx, loc.obs, 64, 200)
    assert a["ok"] == 1 and a["strategy"] == "ranges" and a["cold_frac"] < 0.10
    g = _run(tmp_path, loc.n_cams, loc.lm_off, loc.cam_idx, loc.obs, 64, 200, env={"POVAR_LPL_STRATEGY": "grid"})
    assert g["ok"] == 1 and g["strategy"] == "grid" and g["cold_frac"] > 2 * a["cold_frac"]
    z = _run(tmp_path, zipf.n_cams, zipf.lm_off, zipf.cam_idx, zipf.obs, 64, 200)
    assert z["ok"] == 1 and z["strategy"] == "grid"
    zr = _run(tmp_path, zipf.n_cams, zipf.lm_off, zipf.cam_idx, zipf.obs, 64, 200, env={"POVAR_LPL_STRATEGY": "range"})
    assert zr["ok"] == 1 and zr["strategy"] == "ranges" and zr["cold_frac"] > z["cold_frac"]


@pytest.mark.parametrize("popularity", ["zipf1", "local"])
def test_natural_and_placed_row_orders_share_everything_but_the_rows(tmp_path, popularity):
    """povar_create starts on the natural row order and swaps six arrays when a host thread has placed the rows: tiles,
    workgroup camera sets, partial records and the cold view of the two builds must be identical, and the natural order
    must satisfy the same invariants."""
    from povar_amd import synth
    p = synth.make_problem(300, 20000, 90000, seed=5, popularity=popularity)
    both = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 64, 200, env={"LPL_CHECK_BOTH": "1"})
    assert both["ok"] == 1 and both["placed"] == 1 and both["same_but_rows"] == 1
    nat = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 64, 200, env={"LPL_CHECK_NOPLACE": "1"})
    assert nat["ok"] == 1 and nat["placed"] == 0
    assert nat["rows"] == both["rows"] and nat["tiles"] == both["tiles"] and nat["cold"] == both["cold"]
    assert nat["extra_atomic_lanes_per_half"] > both["extra_atomic_lanes_per_half"]
