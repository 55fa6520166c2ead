"""series_res -- the RESIDENT power series: the whole loop of solve_pOSE (sc/linearization_power_varproj.hpp:191-237:
x_0 = B^-1 (-b), x_i = B^-1 E0 x_{i-1}, early exit :206-229) as ONE launch that keeps rows, landmarks and B^-1 on the chip
(povar_kernels_res.hpp, res_layout.hpp) -- against the CPU oracle at the sizes it is built for (ladybug-49, trafalgar-257,
one rank's landmark shard of venice-1778 at world = 8), against the per-term kernels with robust norms, early exit and
m = 0, and the library's own choice between the two forms.

Tolerances as for the per-term kernels: 20-term increment 1e-10, relative 2-norms.
"""
import os

import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu
ALPHA, LAM, M = 0.01, 1e-4, 20
NT = min(os.cpu_count() or 1, 16)


def _problem(name):
    from povar_amd import capi, synth
    if name.startswith("venice-1778/"):  # rank 0's landmark shard at world = 8 (BASELINE config 4) / 16
        p = synth.make_bal_problem("venice-1778")
        lb, le = capi.shard_range(p.lm_off, int(name.split("/")[1]), 0)
        ob, oe = int(p.lm_off[lb]), int(p.lm_off[le])
        return synth.Problem(p.n_cams, le - lb, (p.lm_off[lb:le + 1] - p.lm_off[lb]).astype(np.int32), p.cam_idx[ob:oe], p.obs[ob:oe],
                             p.cams, p.lms[lb:le])
    return synth.make_bal_problem(name)


def _need_layout(li):
    """The resident layout exists at the sizes of this module -- unless the environment has cut the workgroups it may use
    (tools/forced_mode_suite.sh: POVAR_RES_WGS=7 runs the small problems of the other modules through a seven-workgroup layout)."""
    if li.res_ready != 1 and os.environ.get("POVAR_RES_WGS") is not None:
        pytest.skip("POVAR_RES_WGS leaves no layout at this size")
    assert li.res_ready == 1 and li.res_wgs >= 1, "the resident layout must exist at this size"


def _ctx(p, robust="NONE", **kw):
    from povar_amd import capi
    ctx = capi.Context(p.n_cams, p.lm_off, p.cam_idx, p.obs, robust_norm=robust, e0_mode=capi.E0_IMPLICIT_LDSACC, **kw)
    ctx.layout_finalize(True)
    return ctx


@pytest.mark.parametrize("name", ["ladybug-49", "trafalgar-257", "venice-1778/16", "venice-1778/8"])
def test_resident_series_oracle_parity_at_size(name, monkeypatch):
    """The 20-term increment and the last term of the resident series against the oracle (same linearisation point), and
    against the per-term kernels of the same context.  (The shard of one rank in eight is beyond the size up to which the
    library builds the layout by default -- it is slower than the per-term kernels there --: the limit is lifted.)"""
    from povar_amd import capi
    from oracle import povar_oracle as O
    monkeypatch.setenv("POVAR_RES_MAX_OBS", "1000000")
    p = _problem(name)
    orc = O.Oracle(p.n_cams, p.lm_off, p.cam_idx, p.obs)
    ctx = _ctx(p)
    _need_layout(ctx.layout_info())
    lms = orc.init_landmarks_pose(ALPHA, p.cams)
    ctx.set_cameras(p.cams)
    ctx.set_landmarks(lms)
    assert ctx.linearize_pose(ALPHA)
    st, diag2, jls, sigma, ok = orc.stage1_pose(ALPHA, p.cams, lms)
    assert ok
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, LAM)
    ref, it, status, _ = orc.solve_pose(st, hll, binv, b, M, n_threads=NT)
    ctx.set_series_kernel(0)
    inc0, it0, st0, rc0 = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)
    term0 = ctx.get_term()
    ctx.set_series_kernel(1)
    assert ctx.layout_info().res_active == 1
    for _ in range(2):  # (the second solve replays the captured launch)
        inc, it2, st2, rc = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)
        assert rc == 0 and (it2, st2) == (it, status)
        assert rel(inc, ref) < 1e-10
        assert rel(inc, inc0) < 1e-10 and rel(ctx.get_term(), term0) < 1e-9
    assert ctx.layout_info().res_failed == 0
    del st
    ctx.close()


@pytest.mark.parametrize("robust", ["NONE", "HUBER", "CAUCHY"])
@pytest.mark.parametrize("name", ["ladybug-49", "trafalgar-257"])
def test_resident_series_against_the_per_term_kernels(name, robust):
    """Every robust norm, fixed m, early exit by r_tolerance and by q_tolerance (eta), m = 0 and m = 1: the same increment,
    iteration count and status as the per-term kernels (themselves held against the oracle elsewhere)."""
    from povar_amd import capi
    p = _problem(name)
    ctx = _ctx(p, robust)
    _need_layout(ctx.layout_info())
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    assert ctx.linearize_pose(ALPHA)
    cases = [dict(m=M), dict(m=M, q_tol=0.0, r_tol=1e-3), dict(m=M, q_tol=0.05, r_tol=-1.0), dict(m=40, q_tol=1e-3, r_tol=1e-6),
             dict(m=0), dict(m=1), dict(m=3, q_tol=0.5, r_tol=0.5)]
    want = []
    ctx.set_series_kernel(0)
    for kw in cases:
        want.append(ctx.solve_pose(LAM, capi.POWER_VARPROJ, **kw))
    ctx.set_series_kernel(1)
    for kw, (inc0, it0, st0, rc0) in zip(cases, want):
        inc, it, st, rc = ctx.solve_pose(LAM, capi.POWER_VARPROJ, **kw)
        assert rc == rc0 == 0 and (it, st) == (it0, st0), (kw, it, st, it0, st0)
        assert rel(inc, inc0) < 1e-10, kw
    assert ctx.layout_info().res_failed == 0
    ctx.close()


def test_resident_series_follows_a_new_linearisation_and_damping():
    """The captured launch is replayed across LM iterations: new cameras / landmarks / lambda are picked up (everything the
    kernel keeps on the chip is loaded in its prologue, every launch)."""
    from povar_amd import capi
    p = _problem("trafalgar-257")
    ctx = _ctx(p)
    _need_layout(ctx.layout_info())
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    for lam in (1e-4, 1e-2):
        assert ctx.linearize_pose(ALPHA)
        ctx.set_series_kernel(0)
        inc0 = ctx.solve_pose(lam, capi.POWER_VARPROJ, M)[0]
        ctx.set_series_kernel(1)
        inc = ctx.solve_pose(lam, capi.POWER_VARPROJ, M)[0]
        assert rel(inc, inc0) < 1e-10
        ctx.apply_pose(capi.POWER_VARPROJ, ALPHA, inc0)
    ctx.close()


def test_series_kernel_is_chosen_by_timing_both():
    """Nothing forced: the first series of a context times the per-term kernels and the resident kernel on the prepared
    problem and keeps the faster form; forcing either and handing the choice back work; the increment is the same."""
    from povar_amd import capi
    if os.environ.get("POVAR_RES") is not None:
        pytest.skip("the environment forces the series kernel: this test is about the automatic choice")
    p = _problem("trafalgar-257")
    ctx = _ctx(p)
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    assert ctx.linearize_pose(ALPHA)
    li = ctx.layout_info()
    assert li.res_auto == 1 and li.res_active == 0 and li.res_ready == 1
    inc_auto = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0]
    li = ctx.layout_info()
    assert li.res_auto == 2 and li.tune_terms_us > 0 and li.tune_res_us > 0
    assert (li.res_active == 1) == (li.tune_res_us < 0.98 * li.tune_terms_us)
    for forced in (0, 1):
        ctx.set_series_kernel(forced)
        li = ctx.layout_info()
        assert li.res_auto == 0 and li.res_active == forced
        assert rel(ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0], inc_auto) < 1e-10
    ctx.set_series_kernel(-1)
    assert ctx.layout_info().res_auto == 1
    assert rel(ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0], inc_auto) < 1e-10
    ctx.close()


def test_resident_series_gives_up_and_the_per_term_kernels_take_over(monkeypatch):
    """A launch whose spins run out (here: a spin budget of one poll, so the first hand-over that is not instantly there
    gives up) raises the give-up bit; the library repeats the series with the per-term kernels, keeps the context on them
    and says so -- the caller sees the right increment either way."""
    from povar_amd import capi
    monkeypatch.setenv("POVAR_RES_SPIN", "1")
    p = _problem("trafalgar-257")
    ctx = _ctx(p)
    _need_layout(ctx.layout_info())
    ctx.set_cameras(p.cams)
    ctx.init_landmarks_pose(ALPHA)
    assert ctx.linearize_pose(ALPHA)
    ctx.set_series_kernel(0)
    inc0 = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0]
    ctx.set_series_kernel(1)
    inc = ctx.solve_pose(LAM, capi.POWER_VARPROJ, M)[0]
    assert rel(inc, inc0) < 1e-10
    li = ctx.layout_info()
    if li.res_failed:  # (a device fast enough to have every flag there at the first poll, twenty times in a row, may not)
        assert li.res_active == 0
    ctx.close()
