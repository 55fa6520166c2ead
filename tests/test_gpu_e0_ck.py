"""e0_ck -- the camera-chunk form of right_mul_e0_pOSE (sc/linearization_power_varproj.hpp:364-406; lane = a chunk of one
camera's observations, landmarks of a batch in LDS, povar_kernels_ck.hpp) -- against the CPU oracle at the BASELINE sizes,
every instantiation against e0_lpl, robust norms, and the hand-over between the two kernels when the placed rows (and
with them the chunk layout) arrive from the host thread.

Tolerances as for e0_lpl: E0 x 1e-12, 20-term increment 1e-10, relative 2-norms.
"""
import os

import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu
ALPHA, LAM, M = 0.01, 1e-4, 20
NT = min(os.cpu_count() or 1, 16)
N_VARIANTS = 6


def _problem(name):
    from povar_amd import synth
    if name == "local-900":
        return synth.make_problem(900, 40000, 200000, seed=9, popularity="local")
    return synth.make_bal_problem(name)


def _prepared(p, robust="NONE", **kw):
    from povar_amd import capi
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=robust, e0_mode=capi.E0_IMPLICIT_LDSACC, **kw)
    assert ctx.layout_finalize(True) or p.n_obs < (1 << 20)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    assert ctx.linearize_pose(ALPHA)
    ctx.prepare_pose(LAM)
    return ctx


@pytest.mark.parametrize("name", ["ladybug-49", "trafalgar-257", "venice-1778", "local-900"])
def test_e0_ck_oracle_parity_at_size(name, monkeypatch):
    """E0 x and the 20-term increment of the camera-chunk kernel against the oracle (same linearisation point).
    (ladybug-49 is below the size from which the library uses the lane-per-landmark layout the chunk layout derives
    from: forced, so that BASELINE config 2's shape sees e0_ck too.)"""
    from povar_amd import capi
    from oracle import povar_oracle as O
    if name == "ladybug-49":
        monkeypatch.setenv("POVAR_E0_V1", "0")
        monkeypatch.setenv("POVAR_LPL_PLACE", "sync")
    p = _problem(name)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.layout_finalize(True)
    li = ctx.layout_info()
    assert li.ck_ready == 1 and li.ck_batches >= 1
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    assert ctx.linearize_pose(ALPHA)
    st, diag2, jls, sigma, ok = orc.stage1_pose(ALPHA, p.cams, lms)
    assert ok
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, LAM)
    ctx.prepare_pose(LAM)
    x = np.random.default_rng(5).normal(size=12 * p.n_cams)
    e0_ref = orc.right_mul_e0_pose(st, hll, x, n_threads=NT)
    ref, it, status, _ = orc.solve_pose(st, hll, binv, b, M, n_threads=NT)
    for kernel in (1, 3):
        ctx.set_e0_kernel(kernel)
        assert ctx.layout_info().e0_kernel == kernel
        assert rel(ctx.right_mul_e0_pose(x), e0_ref) < 1e-12, kernel
        inc, it2, st2, rc = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)
        assert rc == 0 and (it2, st2) == (it, status) and rel(inc, ref) < 1e-10, kernel
    del st
    ctx.close()


@pytest.mark.parametrize("robust", ["NONE", "HUBER", "CAUCHY"])
def test_every_e0_ck_instantiation_against_e0_lpl(robust):
    """trafalgar-257 (one landmark batch per workgroup), every instantiation, every robust norm: the same E0 x, the same
    20 terms and the same early exit as the lane-per-landmark kernel (itself held against the oracle elsewhere)."""
    from povar_amd import capi
    p = _problem("trafalgar-257")
    ctx = _prepared(p, robust)
    x = np.random.default_rng(7).normal(size=12 * p.n_cams)
    ctx.set_e0_kernel(0)
    y0 = ctx.right_mul_e0_pose(x)
    inc0, it0, st0, _ = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M, q_tol=0.0, r_tol=1e-3)
    for kernel in range(1, N_VARIANTS + 1):
        ctx.set_e0_kernel(kernel)
        assert rel(ctx.right_mul_e0_pose(x), y0) < 1e-12, kernel
        inc, it, st, rc = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M, q_tol=0.0, r_tol=1e-3)
        assert rc == 0 and (it, st) == (it0, st0) and rel(inc, inc0) < 1e-10, kernel
    ctx.close()


def test_e0_ck_several_batches_and_cold_chunks(monkeypatch):
    """A small LDS budget for cameras (64 accumulator slots) and many landmark batches per workgroup (POVAR_CK_NB=5):
    chunks of cameras without a slot (their own partial records), batches of different sizes, tiles of every height."""
    from povar_amd import capi, synth
    monkeypatch.setenv("POVAR_E0_V1", "0")
    monkeypatch.setenv("POVAR_HOT_ACC", "64")
    monkeypatch.setenv("POVAR_CK_NB", "5")
    monkeypatch.setenv("POVAR_LPL_PLACE", "sync")
    p = synth.make_problem(300, 20000, 90000, seed=5)
    ctx = _prepared(p)
    li = ctx.layout_info()
    assert li.ck_ready == 1 and li.ck_batches >= 5 and li.ck_cold_chunks > 0
    x = np.random.default_rng(3).normal(size=12 * p.n_cams)
    ctx.set_e0_kernel(0)
    y0 = ctx.right_mul_e0_pose(x)
    inc0 = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0]
    for kernel in (1, 3, 4, 6):
        ctx.set_e0_kernel(kernel)
        assert rel(ctx.right_mul_e0_pose(x), y0) < 1e-12
        assert rel(ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0], inc0) < 1e-10
    ctx.close()


def test_e0_ck_on_the_natural_rows_then_on_the_placed_rows(monkeypatch):
    """Rows placed on a host thread (POVAR_LPL_PLACE=async): povar_create builds step 1's chunk layout from the NATURAL rows
    (e0_ck from the first solve on: it does not depend on the order of the lane-per-landmark rows), the chunk layouts of the
    placed rows -- step 2's with them -- replace it when the rows are swapped in.  Same increment before and after (to the
    summation order)."""
    from povar_amd import capi, synth
    monkeypatch.setenv("POVAR_E0_V1", "0")
    monkeypatch.setenv("POVAR_LPL_PLACE", "async")
    monkeypatch.setenv("POVAR_E0_CK", "1")
    p = synth.make_problem(300, 20000, 90000, seed=5)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    before = ctx.layout_info()
    assert before.ck_ready == 1 and before.e0_kernel == 1
    if before.placement == 2:  # not swapped in yet: step 2's layout is still on its way (POVAR_CKH_EARLY=1 builds it too)
        assert before.ckh_ready == 0 and before.e0_kernel_h == 0
    assert ctx.linearize_pose(ALPHA)
    inc_a = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0]
    assert ctx.layout_finalize(True)
    after = ctx.layout_info()
    assert after.placement == 3 and after.ck_ready == 1 and after.e0_kernel == 1 and after.ckh_ready == 1
    assert ctx.linearize_pose(ALPHA)
    inc_b = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0]
    assert rel(inc_b, inc_a) < 1e-10
    ctx.close()


def test_e0_kernel_is_chosen_by_timing_both():
    """Nothing forced: the first power series of a layout times e0_lpl and e0_ck on the prepared problem and keeps the faster
    one; forcing a kernel and handing the choice back both work; the increment is the same either way."""
    from povar_amd import capi
    if os.environ.get("POVAR_E0_CK") is not None or os.environ.get("POVAR_NO_CK") is not None:
        pytest.skip("the environment forces the E0 kernel (tools/forced_mode_suite.sh): this test is about the automatic choice")
    p = _problem("trafalgar-257")
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.layout_finalize(True)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    assert ctx.linearize_pose(ALPHA)
    li = ctx.layout_info()
    assert li.e0_auto == 1 and li.e0_kernel == 0 and li.ck_ready == 1
    inc_auto = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0]
    li = ctx.layout_info()
    assert li.e0_auto == 2 and li.e0_kernel in (0, 1) and li.tune_lpl_us > 0 and li.tune_ck_us > 0
    assert (li.e0_kernel == 1) == (li.tune_ck_us < 0.98 * li.tune_lpl_us)
    for forced in (0, 1):
        ctx.set_e0_kernel(forced)
        li = ctx.layout_info()
        assert li.e0_auto == 0 and li.e0_kernel == forced
        assert rel(ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0], inc_auto) < 1e-10
    ctx.set_e0_kernel(-1)
    assert ctx.layout_info().e0_auto == 1
    assert rel(ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0], inc_auto) < 1e-10
    assert ctx.layout_info().e0_auto == 2
    ctx.close()


@pytest.mark.parametrize("place", ["sync", "async"])
def test_a_layout_the_chunk_kernels_cannot_run_is_not_an_error(monkeypatch, place):
    """More cameras than the camera-chunk kernels' 16-bit ranks hold (here: the limit lowered to ten, POVAR_CK_MAX_CAMS):
    povar_create succeeds on either placement path, no chunk layout is kept, the term loop runs e0_lpl whatever kernel is
    asked for (the synchronous path used to fail the whole context)."""
    from povar_amd import capi, synth
    monkeypatch.setenv("POVAR_E0_V1", "0")
    monkeypatch.setenv("POVAR_CK_MAX_CAMS", "10")
    monkeypatch.setenv("POVAR_LPL_PLACE", place)
    monkeypatch.setenv("POVAR_E0_CK", "1")
    p = synth.make_problem(49, 2000, 8200, seed=7)
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ctx.layout_finalize(True)
    li = ctx.layout_info()
    assert li.ck_ready == 0 and li.e0_kernel == 0 and li.ckh_ready == 0
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    assert ctx.linearize_pose(ALPHA)
    inc, it, st, rc = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)
    assert rc == 0 and it == M and np.all(np.isfinite(inc))
    monkeypatch.delenv("POVAR_CK_MAX_CAMS")
    ref = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, e0_mode=capi.E0_IMPLICIT_LDSACC)
    ref.layout_finalize(True)
    assert ref.layout_info().ck_ready == 1
    ref.set_cameras(p.cams)
    ref.init_landmarks_pose(ALPHA)
    assert ref.linearize_pose(ALPHA)
    assert rel(ref.solve_pose(LAM, capi.POWER_VARPROJ, M)[0], inc) < 1e-10
    ctx.close()
    ref.close()


def test_packed_image_points_are_the_same_operator(monkeypatch):
    """Round 6: where every observation is a six-decimal number (the reference's files: bal_problem.cpp:373-375) the chunk rows
    keep the image points as two int32 of micro-units and the kernel rebuilds the doubles bit for bit (ck_layout.hpp:
    ck_pack_uv verifies every entry).  Packed against POVAR_CK_PACK=0 on the same problem: the same E0 x and the same 20-term
    increment to the summation order of the LDS atomics; a problem with ONE observation that is not such a number keeps the
    16-byte rows and still gives the oracle's answer."""
    from povar_amd import capi, synth
    monkeypatch.setenv("POVAR_E0_V1", "0")
    monkeypatch.setenv("POVAR_LPL_PLACE", "sync")
    p = synth.make_problem(300, 20000, 90000, seed=5)
    x = np.random.default_rng(3).normal(size=12 * p.n_cams)
    out = {}
    for pack in ("1", "0"):
        monkeypatch.setenv("POVAR_CK_PACK", pack)
        for robust in ("NONE", "HUBER"):
            ctx = _prepared(p, robust)
            ctx.set_e0_kernel(1)
            assert ctx.layout_info().ck_packed == int(pack)
            out[pack, robust] = (ctx.right_mul_e0_pose(x), ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0])
            ctx.close()
    for robust in ("NONE", "HUBER"):
        assert rel(out["1", robust][0], out["0", robust][0]) < 1e-14
        assert rel(out["1", robust][1], out["0", robust][1]) < 1e-12
    monkeypatch.delenv("POVAR_CK_PACK")
    obs = p.obs.copy()
    obs[12345, 0] += 1e-9
    import dataclasses
    q = dataclasses.replace(p, obs=obs) if dataclasses.is_dataclass(p) else type(p)(p.n_cams, p.n_lms, p.lm_off, p.cam_idx, obs, p.cams, p.lms)
    ctx = _prepared(q)
    ctx.set_e0_kernel(1)
    assert ctx.layout_info().ck_packed == 0
    y = ctx.right_mul_e0_pose(x)
    ctx.close()
    assert rel(y, out["0", "NONE"][0]) < 1e-9  # (one observation moved by 1e-9 px)


def test_cold_observations_through_the_cold_view_are_the_same_operator(monkeypatch):
    """Round 6: a chunk of a camera WITHOUT an accumulator slot in its workgroup leaves q of each observation (32 bytes) at the
    observation's place in the lane-per-landmark layout's cold camera-major view, and the per-camera kernel forms h~ (x) q as it
    does behind e0_lpl -- instead of a 96-byte partial record per chunk (ck_layout.hpp: cold_q; by itself up to 8 % cold
    observations).  Few accumulator slots (64: a third of the observations cold), several batches, both forms forced in turn:
    the same E0 x, the same 20-term increment; and the same against e0_lpl.  With the HUBER norm the kernel has no cold loop (its
    instantiation has no registers for one) and the library keeps the records whatever the knob says."""
    from povar_amd import capi, synth
    monkeypatch.setenv("POVAR_E0_V1", "0")
    monkeypatch.setenv("POVAR_HOT_ACC", "64")
    monkeypatch.setenv("POVAR_CK_NB", "3")
    monkeypatch.setenv("POVAR_LPL_PLACE", "sync")
    p = synth.make_problem(300, 20000, 90000, seed=5)
    x = np.random.default_rng(3).normal(size=12 * p.n_cams)
    out = {}
    for form, env in (("cold view", {"POVAR_CK_COLD_Q_ALWAYS": "1"}), ("records", {"POVAR_CK_COLD_RECORDS": "1"})):
        for k in ("POVAR_CK_COLD_Q_ALWAYS", "POVAR_CK_COLD_RECORDS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for robust in ("NONE", "HUBER"):
            ctx = _prepared(p, robust)
            li = ctx.layout_info()
            assert li.ck_ready == 1 and li.ck_cold_chunks > 1000
            assert li.ck_cold_q == (1 if form == "cold view" and robust == "NONE" else 0)
            ctx.set_e0_kernel(0)
            y_lpl = ctx.right_mul_e0_pose(x)
            for kernel in (1, 3, 4):
                ctx.set_e0_kernel(kernel)
                assert rel(ctx.right_mul_e0_pose(x), y_lpl) < 1e-12, (form, robust, kernel)
            ctx.set_e0_kernel(1)
            out[form, robust] = (ctx.right_mul_e0_pose(x), ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0])
            ctx.close()
    for robust in ("NONE", "HUBER"):
        assert rel(out["cold view", robust][0], out["records", robust][0]) < 1e-13
        assert rel(out["cold view", robust][1], out["records", robust][1]) < 1e-11


def test_step2_wide_stride_with_capped_accumulators_is_the_same_operator(monkeypatch):
    """Round 6: e0_ck_h with LDS arrays cut for 2048 landmark slots instead of 1536 where that saves a landmark batch (venice:
    three batches -> two) -- then only 314 accumulators fit beside them: a workgroup keeps those of its most observed cameras,
    the chunks of the other cameras get records of their own (ck_layout.hpp: CkShape::wide_slots).  700 cameras over 12
    workgroups (one batch of 2048 slots against two of 1536; the parent layout has more than 314 camera slots per workgroup):
    the library takes the wide stride by itself, and the 20-term step-2 increment is the same with either stride forced and
    with e0_lpl_h, NONE and HUBER."""
    from povar_amd import capi, synth
    monkeypatch.setenv("POVAR_E0_V1", "0")
    monkeypatch.setenv("POVAR_E0_WGS", "12")
    monkeypatch.setenv("POVAR_LPL_PLACE", "sync")
    for k in ("POVAR_CKH_ACC_CAP", "POVAR_HOT_ACC", "POVAR_CK_NB", "POVAR_LPL_K0", "POVAR_LPL_STRATEGY"):  # (tools/forced_mode_suite.sh)
        monkeypatch.delenv(k, raising=False)
    p = synth.make_problem(700, 20000, 90000, seed=6)
    rng = np.random.default_rng(11)
    cams = rng.normal(size=(p.n_cams, 12))
    cams[:, 8:11] *= 0.1
    cams[:, 11] = 5 + rng.random(p.n_cams)
    cams /= np.linalg.norm(cams, axis=1, keepdims=True)
    lms_h = np.concatenate([rng.normal(size=(p.n_lms, 3)), np.ones((p.n_lms, 1))], 1)
    obs = p.obs / 500.0
    out = {}
    for stride in (None, "1536", "2048"):
        monkeypatch.delenv("POVAR_CKH_STRIDE", raising=False)
        if stride:
            monkeypatch.setenv("POVAR_CKH_STRIDE", stride)
        for robust in ("NONE", "HUBER"):
            ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, obs, robust_norm=robust, huber=0.5, e0_mode=capi.E0_IMPLICIT_LDSACC)
            li = ctx.layout_info()
            assert li.ckh_ready == 1 and li.ckh_stride == int(stride or 2048) and li.ckh_slots <= li.ckh_stride
            if li.ckh_stride == 2048:
                assert li.ckh_batches == 1 and li.ckh_accumulators == 314 < li.lds_slots and li.ckh_capped_obs > 0
            else:
                assert li.ckh_batches == 2 and li.ckh_accumulators == li.lds_slots and li.ckh_capped_obs == 0
            ctx.set_cameras(cams)
            ctx.set_landmarks_homogeneous(lms_h)
            assert ctx.linearize_homogeneous()
            ctx.prepare_joint(LAM)
            res = {}
            for kernel in (0, 1):
                ctx.set_e0_kernel(kernel)
                assert ctx.layout_info().e0_kernel_h == kernel
                ctx.power_series_begin()
                ctx.power_series_step()
                t1 = ctx.get_term(11).copy()
                inc, it, st, rc = ctx.solve_joint(LAM, M)
                assert rc == 0 and it == M
                res[kernel] = (t1, inc)
            assert rel(res[1][0], res[0][0]) < 1e-12 and rel(res[1][1], res[0][1]) < 1e-10, (stride, robust)
            out[stride, robust] = res[1]
            ctx.close()
    for robust in ("NONE", "HUBER"):
        assert rel(out["2048", robust][0], out["1536", robust][0]) < 1e-12
        assert rel(out["2048", robust][1], out["1536", robust][1]) < 1e-10
        assert np.array_equal(out[None, robust][1], out["2048", robust][1]) or rel(out[None, robust][1], out["2048", robust][1]) < 1e-12
