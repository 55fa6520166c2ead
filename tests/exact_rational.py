"""Exact rational arithmetic for small step-1 problems, straight from the REFERENCE's formulas (not from either restatement):
test infrastructure that holds the C oracle and the HIP path to the mathematically exact E0 x and b.

Inputs are IEEE doubles, i.e. exact rationals; everything below is `fractions.Fraction` arithmetic, no rounding anywhere.

    residual / Jacobians   bal/bal_bundle_adjustment_helper.cpp:251-310
        M = [ sb (P0 - u P2); sb (P1 - v P2); sa P0; sa P1 ]  (rows of the 3x4 camera P),  sa = sqrt(alpha), sb = sqrt(1 - alpha)
        res = M [x; 1] - (0, 0, sa u, sa v);   Jl = M[:, :3];   Jp rows: sb [h, 0, -u h], sb [0, h, -v h], sa [h, 0, 0], sa [0, h, 0], h = [x; 1]
    Hll^-1, b              sc/landmark_block.hpp:517-536     Hll = Jl^T Jl;  w = Hll^-1 Jl^T r;  b_c += Jp_i^T (r_i - Jl_i w)
    E0 x                   sc/linearization_power_varproj.hpp:377-396   y_c += Jp_i^T [ Jl Hll^-1 Jl^T (Jp x) ]_i

sa and sb are irrational, but every quantity above is a sum over residual ROWS of products of two factors that carry the same
one of them: with Jp = D Jp0, Jl = D Jl0, r = D r0, D = diag(sb, sb, sa, sa), only D^2 = diag(1 - alpha, 1 - alpha, alpha, alpha)
appears -- rational.  The column scalings the linearizor applies (Jl diag(s), Jp diag(sigma): landmark_block.hpp:284-295, 324-334)
are irrational too; s cancels exactly (Jl S (S Hll S)^-1 S Jl^T = Jl Hll^-1 Jl^T), and sigma is handled by the caller:
E0_scaled x = sigma * E0 (sigma * x), b_scaled = sigma * b with the sigma DOUBLES of the implementation under test taken as exact
numbers (sigma itself is checked against a 60-digit evaluation of 1 / (eps + sqrt(d)), linearizor_power_varproj.cpp:68-70)."""
from decimal import Decimal, getcontext
from fractions import Fraction as F


def _rows(P, x, u, v):
    """Jp0 (4 x 12), Jl0 (4 x 3), r0 (4) of one observation WITHOUT the factors sb / sa (bal_bundle_adjustment_helper.cpp:251-310)."""
    P0, P1, P2 = P[0:4], P[4:8], P[8:12]
    h = [x[0], x[1], x[2], F(1)]
    M = [[P0[j] - u * P2[j] for j in range(4)], [P1[j] - v * P2[j] for j in range(4)], list(P0), list(P1)]
    r0 = [sum(M[r][j] * h[j] for j in range(4)) for r in range(4)]
    r0[2] -= u
    r0[3] -= v
    z4 = [F(0)] * 4
    Jp0 = [h + z4 + [-u * t for t in h], z4 + h + [-v * t for t in h], h + z4 + z4, z4 + h + z4]
    Jl0 = [M[r][:3] for r in range(4)]
    return Jp0, Jl0, r0


def _inv3(H):
    a, b, c, d, e, f, g, h, i = H[0][0], H[0][1], H[0][2], H[1][0], H[1][1], H[1][2], H[2][0], H[2][1], H[2][2]
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    adj = [[e * i - f * h, c * h - b * i, b * f - c * e], [f * g - d * i, a * i - c * g, c * d - a * f], [d * h - e * g, b * g - a * h, a * e - b * d]]
    return [[adj[r][s] / det for s in range(3)] for r in range(3)]


class ExactStep1:
    """E0 x, b and d = diag(Jp^T Jp) of the UNSCALED system in exact arithmetic."""

    def __init__(self, alpha, n_cams, lm_off, cam_idx, obs, cams, lms):
        self.n_cams = int(n_cams)
        al = F(float(alpha))
        self.D2 = [1 - al, 1 - al, al, al]
        self.lm = []
        for l in range(len(lm_off) - 1):
            x = [F(float(t)) for t in lms[l]]
            ob = []
            for i in range(int(lm_off[l]), int(lm_off[l + 1])):
                c = int(cam_idx[i])
                P = [F(float(t)) for t in cams[c]]
                ob.append((c,) + _rows(P, x, F(float(obs[i][0])), F(float(obs[i][1]))))
            H = [[sum(self.D2[r] * Jl0[r][a] * Jl0[r][b] for _, _, Jl0, _ in ob for r in range(4)) for b in range(3)] for a in range(3)]
            self.lm.append((ob, _inv3(H)))

    def diag2(self):
        d = [F(0)] * (12 * self.n_cams)
        for ob, _ in self.lm:
            for c, Jp0, _, _ in ob:
                for j in range(12):
                    d[12 * c + j] += sum(self.D2[r] * Jp0[r][j] ** 2 for r in range(4))
        return d

    def e0(self, x):
        """E0 x for a vector of doubles or Fractions (linearization_power_varproj.hpp:377-396)."""
        xs = [t if isinstance(t, F) else F(float(t)) for t in x]
        y = [F(0)] * (12 * self.n_cams)
        for ob, Hi in self.lm:
            u = [F(0)] * 3
            for c, Jp0, Jl0, _ in ob:
                t = [self.D2[r] * sum(Jp0[r][j] * xs[12 * c + j] for j in range(12)) for r in range(4)]  # D^2 (Jp0 x)
                for a in range(3):
                    u[a] += sum(Jl0[r][a] * t[r] for r in range(4))
            g = [sum(Hi[a][b] * u[b] for b in range(3)) for a in range(3)]
            for c, Jp0, Jl0, _ in ob:
                s = [self.D2[r] * sum(Jl0[r][a] * g[a] for a in range(3)) for r in range(4)]
                for j in range(12):
                    y[12 * c + j] += sum(Jp0[r][j] * s[r] for r in range(4))
        return y

    def b(self):
        """b (landmark_block.hpp:517-536)."""
        out = [F(0)] * (12 * self.n_cams)
        for ob, Hi in self.lm:
            g = [sum(self.D2[r] * Jl0[r][a] * r0[r] for _, _, Jl0, r0 in ob for r in range(4)) for a in range(3)]
            w = [sum(Hi[a][b] * g[b] for b in range(3)) for a in range(3)]
            for c, Jp0, Jl0, r0 in ob:
                s = [self.D2[r] * (r0[r] - sum(Jl0[r][a] * w[a] for a in range(3))) for r in range(4)]
                for j in range(12):
                    out[12 * c + j] += sum(Jp0[r][j] * s[r] for r in range(4))
        return out


def sigma_60_digits(d, eps):
    """1 / (eps + sqrt(d)) (linearizor_power_varproj.cpp:68-70) to 60 digits, returned as floats (correctly rounded)."""
    getcontext().prec = 60
    e = Decimal(float(eps))
    return [float(1 / (e + (Decimal(t.numerator) / Decimal(t.denominator)).sqrt())) for t in d]


def rel_err(approx, exact):
    """|| approx - exact || / || exact || with the difference taken exactly."""
    num = sum((F(float(a)) - e) ** 2 for a, e in zip(approx, exact))
    den = sum(e ** 2 for e in exact)
    return float(num / den) ** 0.5
