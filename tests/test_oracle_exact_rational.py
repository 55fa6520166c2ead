"""The C oracle against EXACT rational arithmetic on the small golden problems: E0 x, b and the pose
scaling evaluated straight from the reference's formulas (tests/exact_rational.py cites them) with fractions.Fraction -- no
rounding, no NumPy, nothing shared with either restatement except the input doubles.  It cannot lift "parity unpinned" (the
reference still pins nothing), but it takes "two restatements sharing one misreading of the arithmetic" off the list: a wrong
sign, index, weight or scaling shows at 1e-3, the bar here is 1e-13."""
import os

import numpy as np

from exact_rational import ExactStep1, rel_err, sigma_60_digits

HERE = os.path.dirname(os.path.abspath(__file__))


def _oracle_state(g):
    from oracle import povar_oracle as O
    orc = O.Oracle(int(g["n_cams"]), g["lm_off"], g["cam_idx"], g["obs"])
    st, diag2, jls, sigma, ok = orc.stage1_pose(float(g["alpha"]), g["cams"], g["lms"])
    assert ok
    orc.scale_jp_cols_pose(st, sigma)
    hll, b, binv = orc.prepare_hb_pose(st, float(g["lam"]))
    return orc, st, hll, b, diag2, sigma


def test_oracle_e0_b_and_scaling_against_exact_rational_arithmetic():
    g = np.load(os.path.join(HERE, "golden", "step1_small_none.npz"))
    ex = ExactStep1(float(g["alpha"]), int(g["n_cams"]), g["lm_off"], g["cam_idx"], g["obs"], g["cams"], g["lms"])
    orc, st, hll, b, diag2, sigma = _oracle_state(g)
    # d = diag(Jp^T Jp) (landmark_block.hpp:272-282) and sigma = 1 / (eps + sqrt(d)) (linearizor_power_varproj.cpp:68-70)
    d = ex.diag2()
    assert rel_err(diag2, d) < 1e-14
    s60 = np.array(sigma_60_digits(d, float(g["eps"])))
    assert np.abs(sigma / s60 - 1).max() < 1e-14
    # E0_scaled x = sigma * E0 (sigma * x) and b_scaled = sigma * b, with the oracle's own sigma doubles as exact numbers
    from fractions import Fraction as F
    sg = [F(float(t)) for t in sigma]
    rng = np.random.default_rng(5)
    for trial in range(2):
        x = rng.normal(size=12 * int(g["n_cams"]))
        sx = [F(float(a)) * s for a, s in zip(x, sg)]
        y_exact = [s * t for s, t in zip(sg, ex.e0(sx))]
        y = orc.right_mul_e0_pose(st, hll, x)
        assert rel_err(y, y_exact) < 1e-13, rel_err(y, y_exact)
    b_exact = [s * t for s, t in zip(sg, ex.b())]
    assert rel_err(b, b_exact) < 1e-13, rel_err(b, b_exact)
    # and the committed golden vectors (NumPy restatement): the same bar against the exact values
    assert rel_err(g["b"], b_exact) < 1e-12 and rel_err(g["diag2"], d) < 1e-13
