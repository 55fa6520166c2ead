"""Host-only invariants of the camera-chunk layout builder (povar_amd/csrc/ck_layout.hpp: the layout of e0_ck, derived
from the lane-per-landmark layout) through tests/cpp/ck_layout_check.cpp: every observation in exactly one chunk of its
camera, landmark slots that name its landmark inside the batch, accumulator slots / own partial records consistent with
the workgroup's camera set, partial records camera-major and used once, LDS capacity respected.  Runs without a GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "ck_layout_check")


def _run(tmp_path, n_cams, lm_off, cam_idx, obs, grid, n_acc, n_waves=16, env=None):
    src = [os.path.join(ROOT, "tests", "cpp", "ck_layout_check.cpp"), os.path.join(ROOT, "povar_amd", "csrc", "ck_layout.hpp"),
           os.path.join(ROOT, "povar_amd", "csrc", "lpl_layout.hpp")]
    if not os.path.exists(BIN) or any(os.path.getmtime(BIN) < os.path.getmtime(s) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-B", "../../build/ck_layout_check"],
                              stdout=subprocess.DEVNULL)
    f = [str(tmp_path / n) for n in ("lm_off.bin", "cam_idx.bin", "obs.bin")]
    np.ascontiguousarray(lm_off, dtype=np.int32).tofile(f[0])
    np.ascontiguousarray(cam_idx, dtype=np.int32).tofile(f[1])
    np.ascontiguousarray(obs, dtype=np.float64).tofile(f[2])
    r = subprocess.run([BIN, str(n_cams)] + f + [str(grid), str(n_acc), str(n_waves)], capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("grid,n_acc,n_waves", [(256, 520, 16), (256, 64, 16), (7, 40, 8), (1, 520, 16), (12, 16, 12)])
def test_ck_layout_invariants_medium(tmp_path, grid, n_acc, n_waves):
    from povar_amd import synth
    p = synth.make_problem(300, 20000, 90000, seed=5)
    # (the generator's observations are six-decimal numbers, as the reference's files hold them: the rows are packed)
    # (... and cold lanes leave q in the parent's cold view: forced here whatever the cold share, so that the invariants of that
    # form are checked on layouts with few accumulator slots too; by itself the builder takes it up to 8 % cold observations)
    s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, grid, n_acc, n_waves, env={"CK_CHECK_WANT_PACKED": "1", "POVAR_CK_COLD_Q_ALWAYS": "1"})
    assert s["ok"] == 1 and s["lds_bytes"] <= 160 * 1024 and s["packed"] == 1 and s["cold_q"] == 1
    auto = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, grid, n_acc, n_waves)
    assert auto["ok"] == 1 and auto["cold_q"] == (1 if 100 * s["cold_obs"] <= 8 * p.n_obs else 0)
    if p.n_cams <= n_acc:
        assert s["cold_chunks"] == 0
    # ... and the form with a partial record per cold chunk (e0_ck_det's layout, POVAR_CK_COLD_RECORDS=1)
    r = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, grid, n_acc, n_waves, env={"CK_CHECK_COLD_RECORDS": "1"})
    assert r["ok"] == 1 and r["cold_q"] == 0 and r["cold_chunks"] == s["cold_chunks"] and r["part_rec"] == s["part_rec"] + s["cold_chunks"]


def test_ck_layout_packs_the_image_points_only_when_every_one_comes_back(tmp_path):
    """Packed rows (two int32 of micro-units per observation) exist only where EVERY image point is RN(k / 10^6) with
    |k| < 2^31 and the device's decode sequence returns its bits (the checker decodes every row and compares the bits with the
    problem's): arbitrary doubles, one observation off by an ulp, a coordinate beyond 2147 px or POVAR_CK_PACK=0 keep the
    16-byte rows."""
    from povar_amd import synth
    p = synth.make_problem(60, 3000, 14000, seed=11)
    assert _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 16, 60, env={"CK_CHECK_WANT_PACKED": "1"})["packed"] == 1
    assert _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 16, 60, env={"CK_CHECK_WANT_PACKED": "0", "POVAR_CK_PACK": "0"})["packed"] == 0
    obs = p.obs.copy()
    obs[777, 1] = np.nextafter(obs[777, 1], np.inf)
    assert _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, obs, 16, 60, env={"CK_CHECK_WANT_PACKED": "0"})["packed"] == 0
    obs = p.obs.copy()
    obs[5, 0] = 2147.483648
    assert _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, obs, 16, 60, env={"CK_CHECK_WANT_PACKED": "0"})["packed"] == 0
    obs = p.obs + np.random.default_rng(1).normal(size=p.obs.shape) * 1e-9
    assert _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, obs, 16, 60, env={"CK_CHECK_WANT_PACKED": "0"})["packed"] == 0


def test_ck_layout_long_tracks_and_single_observation_landmarks(tmp_path):
    """Landmarks dealt over several lanes (one landmark slot for all of them), 1-observation landmarks, a 650-observation
    track, natural and placed row orders of the parent layout."""
    rng = np.random.default_rng(3)
    n_c = 700
    ks = np.concatenate([[1, 1, 2, 650, 130, 64, 65, 9, 8], rng.integers(1, 12, size=3000)])
    lm_off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    w = 1.0 / np.arange(1, n_c + 1)
    cam_idx = np.concatenate([np.sort(rng.choice(n_c, k, replace=False, p=w / w.sum())) for k in ks]).astype(np.int32)
    obs = rng.normal(size=(cam_idx.shape[0], 2))
    for env in (None, {"LPL_CHECK_NOPLACE": "1"}, {"CK_CHECK_NOPLACE": "1"}):
        s = _run(tmp_path, n_c, lm_off, cam_idx, obs, 16, 100, env=env)
        assert s["ok"] == 1


def test_ck_layout_venice_shape(tmp_path):
    """The BASELINE shape: two landmark batches per workgroup, chunks of 7-9 observations on average, a few per cent of
    padding, the bank placement of the rows at most one extra lane per row half."""
    from povar_amd import synth
    p = synth.make_bal_problem("venice-1778")
    s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 256, 520, env={"LPL_CHECK_NOPLACE": "1"})
    assert s["ok"] == 1 and s["nb"] == 2 and s["pad_frac"] < 0.06 and s["obs_per_chunk"] > 6.5
    assert s["extra_lanes_per_half_row"] < 1.1


def test_ck_layout_step2_shape(tmp_path):
    """The instance e0_ck_h (step 2) runs: 64 bytes of LDS per landmark slot in component-major arrays with a compile-time
    stride -- 1536 slots, or 2048 where that saves a landmark batch; then the accumulators that no longer fit beside the slots
    are given up: a workgroup keeps those of its most observed cameras, the other cameras' chunks get records of their own
    (the checker: every accumulator serves one camera, a camera with one has no cold chunk in that workgroup, every record is
    used once, the capped observations are counted)."""
    from povar_amd import synth
    p = synth.make_problem(300, 20000, 90000, seed=5)
    for grid, n_acc in ((256, 520), (7, 40), (1, 520)):
        s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, grid, n_acc, env={"CK_CHECK_STEP2": "1"})
        assert s["ok"] == 1 and s["slots"] <= s["stride"] and s["lds_bytes"] <= 160 * 1024
    # 20 000 landmarks over 12 workgroups: 31 lane tiles each -- two batches of 1536 slots or one of 2048; 300 cameras fit beside either
    s = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 12, 520, env={"CK_CHECK_STEP2": "1"})
    assert s["ok"] == 1 and s["stride"] == 2048 and s["nb"] == 1 and s["capped_obs"] == 0 and s["max_acc"] <= 314
    f = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 12, 520, env={"CK_CHECK_STEP2": "1", "POVAR_CKH_STRIDE": "1536"})
    assert f["ok"] == 1 and f["stride"] == 1536 and f["nb"] == 2
    # ... and with 700 cameras (501 slots in the parent layout at most) the wide stride caps the accumulators at 314
    q = synth.make_problem(700, 20000, 90000, seed=6)
    s = _run(tmp_path, q.n_cams, q.lm_off, q.cam_idx, q.obs, 12, 520, env={"CK_CHECK_STEP2": "1"})
    assert s["ok"] == 1 and s["stride"] == 2048 and s["nb"] == 1 and s["max_acc"] == 314 and s["capped_obs"] > 0
    f = _run(tmp_path, q.n_cams, q.lm_off, q.cam_idx, q.obs, 12, 520, env={"CK_CHECK_STEP2": "1", "POVAR_CKH_STRIDE": "1536"})
    assert f["ok"] == 1 and f["capped_obs"] == 0 and f["max_acc"] > 314 and f["cold_chunks"] < s["cold_chunks"]
    p = synth.make_bal_problem("venice-1778")
    s1 = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 256, 520, env={"LPL_CHECK_NOPLACE": "1"})
    s2 = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 256, 520, env={"LPL_CHECK_NOPLACE": "1", "CK_CHECK_STEP2": "1"})
    assert s2["ok"] == 1 and s2["nb"] == 2 == s1["nb"] and s2["stride"] == 2048 and s2["max_acc"] == 314 and s2["obs_per_chunk"] > 4.5
    s3 = _run(tmp_path, p.n_cams, p.lm_off, p.cam_idx, p.obs, 256, 520, env={"LPL_CHECK_NOPLACE": "1", "CK_CHECK_STEP2": "1", "POVAR_CKH_STRIDE": "1536"})
    assert s3["ok"] == 1 and s3["nb"] == 3 and s3["slots"] <= 1536 and s3["capped_obs"] == 0
