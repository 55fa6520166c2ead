// povar_kernels_joint.hpp -- step 2 (projective refinement on the Riemannian manifold, RIPOBA):
// the same kernel structure as step 1 with 2-row tiles.
//
// Stored step-2 tile of one observation (landmark_block.hpp:180-269 after scale_Jl_cols_homogeneous
// :298-309, scale_Jp_cols_joint :311-321 and linearize_nullspace :227-269), with pc = P X = (x,y,z),
// D = [[1/z, 0, -x/z^2], [0, 1/z, -y/z^2]] (bal_camera.hpp:153-160):
//     Jp12 = sw * [ D00 X, 0, D02 X ; 0, D00 X, D12 X ] * diag(sigma_c)      (helper.cpp:352-369)
//     Jl4  = sw * (D P) * diag(s_l)                                           (helper.cpp:376)
//     Jp11 = Jp12 N_c,  Jl3 = Jl4 N_l,  r = sw * (x/z - u, y/z - v)
// N_c (12x11) and N_l (4x3) are orthonormal bases of null(vec(P_c)^T) and null(X_l^T).  The
// reference takes them from Eigen's CompleteOrthogonalDecomposition (helper.cpp:202-216); every
// orthonormal basis gives the same ambient increments, norms and convergence tests (A.7), so one
// Householder reflector is used: N = H[:, 1:], H = I - beta w w^T, w = v + sign(v0)|v| e0,
// beta = 2 / w^T w.  Products with N are rank-one updates:
//     (a N)_j = a_{j+1} - beta (a.w) w_{j+1}        (N x)_i = [0;x]_i - beta w_i (w[1:].x)
#pragma once
#include "povar_kernels.hpp"

namespace povar {

struct Hom {
  double D00, D02, D12, r0, r1;
  bool valid;
};
__device__ inline Hom hom_project(const Cam& P, const double4& X, double u, double v) {
  const double x = dot4(P.r0, X), y = dot4(P.r1, X), z = dot4(P.r2, X);
  Hom h;
  h.r0 = x / z - u;
  h.r1 = y / z - v;
  h.D00 = 1 / z;
  h.D02 = -x / (z * z);
  h.D12 = -y / (z * z);
  h.valid = fabs(z) >= 1e-5;  // bal_camera.hpp:147 (Sophus epsilonSqrt<double>)
  return h;
}
// Jl4 = scale * (D P) * diag(s): row0 = D00 P0 + D02 P2, row1 = D00 P1 + D12 P2
__device__ inline void hom_jl4(const Cam& P, const Hom& h, double scale, const double4& s, double (&jl)[8]) {
  jl[0] = scale * (h.D00 * P.r0.x + h.D02 * P.r2.x) * s.x;
  jl[1] = scale * (h.D00 * P.r0.y + h.D02 * P.r2.y) * s.y;
  jl[2] = scale * (h.D00 * P.r0.z + h.D02 * P.r2.z) * s.z;
  jl[3] = scale * (h.D00 * P.r0.w + h.D02 * P.r2.w) * s.w;
  jl[4] = scale * (h.D00 * P.r1.x + h.D12 * P.r2.x) * s.x;
  jl[5] = scale * (h.D00 * P.r1.y + h.D12 * P.r2.y) * s.y;
  jl[6] = scale * (h.D00 * P.r1.z + h.D12 * P.r2.z) * s.z;
  jl[7] = scale * (h.D00 * P.r1.w + h.D12 * P.r2.w) * s.w;
}
// Householder vector of a 4-vector (kernel_COD of X^T, helper.cpp:202-216)
__device__ inline void house4(const double4& X, double (&w)[4], double& beta) {
  const double nv = sqrt(X.x * X.x + X.y * X.y + X.z * X.z + X.w * X.w);
  w[0] = X.x + (X.x >= 0 ? nv : -nv);
  w[1] = X.y;
  w[2] = X.z;
  w[3] = X.w;
  beta = 2.0 / (w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + w[3] * w[3]);
}
__device__ inline void jl3_of_jl4(const double (&jl4)[8], const double (&w)[4], double beta, double (&jl3)[6]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const double aw = jl4[4 * r] * w[0] + jl4[4 * r + 1] * w[1] + jl4[4 * r + 2] * w[2] + jl4[4 * r + 3] * w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) jl3[3 * r + j] = jl4[4 * r + j + 1] - beta * aw * w[j + 1];
  }
}
// t = Jp12 * p for the structured Jp12, zc = (sigma * p)[12c..]
__device__ inline void hom_jp_x(const Hom& h, const double4& X, double scale, const double4* zc, double (&t)[2]) {
  const double d0 = dot4(X, zc[0]), d1 = dot4(X, zc[1]), d2 = dot4(X, zc[2]);
  t[0] = scale * (h.D00 * d0 + h.D02 * d2);
  t[1] = scale * (h.D00 * d1 + h.D12 * d2);
}
__device__ inline double4 hom_q(const Hom& h, double scale, double s0, double s1) {
  return make_double4(scale * h.D00 * s0, scale * h.D00 * s1, scale * (h.D02 * s0 + h.D12 * s1), scale);
}
__device__ inline void acc_h6(double* red, const double (&jl3)[6]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    red[0] += jl3[3 * r] * jl3[3 * r];
    red[1] += jl3[3 * r] * jl3[3 * r + 1];
    red[2] += jl3[3 * r] * jl3[3 * r + 2];
    red[3] += jl3[3 * r + 1] * jl3[3 * r + 1];
    red[4] += jl3[3 * r + 1] * jl3[3 * r + 2];
    red[5] += jl3[3 * r + 2] * jl3[3 * r + 2];
  }
}

// K2': compute_error_projective_space_homogeneous (helper.cpp:157-196)
struct OpErrorH {
  static constexpr int NRED = 0, NSC = 6;
  static constexpr bool CHECK_DONE = false;
  using Local = NoLocal;
  __device__ void phase1(const Dp&, int, int, int, double2, Local&, double*) const {}
  __device__ void phase2(const Dp& d, int, int cam, int lm, double2 uv, Local&, const double*, double* sc) const {
    const Hom h = hom_project(load_cam(d.cams4, cam), d.lms4[lm], uv.x, uv.y);
    const double r2 = h.r0 * h.r0 + h.r1 * h.r1;
    if (!isfinite(r2)) atomicOr(&d.flags[0], 1);
    double e, w;
    error_weight(d, r2, e, w);
    sc[0] += e; sc[1] += sqrt(r2); sc[2] += 1.0;
    if (h.valid) { sc[3] += e; sc[4] += sqrt(r2); sc[5] += 1.0; }
  }
  __device__ void finish_lm(const Dp&, int, const double*) const {}
};

// K3' + K5': linearize_landmark_projective_space_homogeneous (landmark_block.hpp:180-225) and
// scale_Jl_cols_homogeneous (:298-309)
struct OpLinearizeH {
  static constexpr int NRED = 4, NSC = 0;
  static constexpr bool CHECK_DONE = false;
  using Local = NoLocal;
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local&, double* red) const {
    const Cam P = load_cam(d.cams_lin4, cam);
    const Hom h = hom_project(P, d.lms_lin4[lm], uv.x, uv.y);
    const double r2 = h.r0 * h.r0 + h.r1 * h.r1;
    double e, w;
    error_weight(d, r2, e, w);
    const double sw = sqrt(w);
    if (!isfinite(r2) || !isfinite(sw) || !isfinite(h.D02) || !isfinite(h.D12)) atomicOr(&d.flags[0], 1);
    d.sw[slot] = sw;
    if (d.robust && d.v2.w && !d.lin_aux_only) d.v2.w[d.v2.of_slot[slot]] = w;
    d.rres[slot] = make_double4(sw * h.r0, sw * h.r1, 0, 0);
    double jl[8];
    hom_jl4(P, h, sw, make_double4(1, 1, 1, 1), jl);
#pragma unroll
    for (int j = 0; j < 4; ++j) red[j] += jl[j] * jl[j] + jl[4 + j] * jl[4 + j];
  }
  __device__ void phase2(const Dp&, int, int, int, double2, Local&, const double*, double*) const {}
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    if (d.lin_aux_only) return;  // only the per-slot arrays are wanted (ensure_legacy, povar_lm.hip)
    d.jl_scale4[lm] = make_double4(1.0 / (d.eps + sqrt(tot[0])), 1.0 / (d.eps + sqrt(tot[1])),
                                   1.0 / (d.eps + sqrt(tot[2])), 1.0 / (d.eps + sqrt(tot[3])));
  }
};

struct HomObs {  // everything an observation needs from the linearisation point
  Hom h;
  double jl4[8], jl3[6], sw;
  double4 X;
  __device__ inline void load(const Dp& d, int slot, int cam, int lm, double2 uv) {
    const Cam P = load_cam(d.cams_lin4, cam);
    X = d.lms_lin4[lm];
    h = hom_project(P, X, uv.x, uv.y);
    sw = d.robust ? d.sw[slot] : 1.0;
    hom_jl4(P, h, sw, d.jl_scale4[lm], jl4);
    double w[4], beta;
    house4(X, w, beta);
    jl3_of_jl4(jl4, w, beta, jl3);
  }
};

__device__ inline void hinv_damped(const double* tot, double lambda, double (&Hi)[9]) {
  double H[9];
  sym3(tot, H);
  H[0] += lambda; H[4] += lambda; H[8] += lambda;  // Proj^T lambda Proj = lambda I_3 (landmark_block.hpp:485)
  inv3(H, Hi);
}

// K7' (landmark part): get_Hll_inv_add_Hpp_b_joint (landmark_block.hpp:474-507)
struct OpPrepareH {
  static constexpr int NRED = 9, NSC = 0;
  static constexpr bool CHECK_DONE = false;
  struct Local { HomObs o; double4 r; };
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local& L, double* red) const {
    L.o.load(d, slot, cam, lm, uv);
    L.r = d.rres[slot];
    acc_h6(red, L.o.jl3);
#pragma unroll
    for (int j = 0; j < 3; ++j) red[6 + j] += L.o.jl3[j] * L.r.x + L.o.jl3[3 + j] * L.r.y;
  }
  __device__ void phase2(const Dp& d, int slot, int, int, double2, Local& L, const double* tot, double*) const {
    double Hi[9];
    hinv_damped(tot, d.lambda_lm, Hi);
    const double w0 = Hi[0] * tot[6] + Hi[1] * tot[7] + Hi[2] * tot[8];
    const double w1 = Hi[3] * tot[6] + Hi[4] * tot[7] + Hi[5] * tot[8];
    const double w2 = Hi[6] * tot[6] + Hi[7] * tot[7] + Hi[8] * tot[8];
    const double e0 = L.r.x - (L.o.jl3[0] * w0 + L.o.jl3[1] * w1 + L.o.jl3[2] * w2);
    const double e1 = L.r.y - (L.o.jl3[3] * w0 + L.o.jl3[4] * w1 + L.o.jl3[5] * w2);
    d.q4[slot] = hom_q(L.o.h, L.o.sw, e0, e1);
  }
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    double Hi[9];
    hinv_damped(tot, d.lambda_lm, Hi);
#pragma unroll
    for (int k = 0; k < 9; ++k) d.hll_inv[9 * (size_t)lm + k] = Hi[k];
    // packed 128-byte landmark record of the per-term kernel: X | s | Hll^-1 (symmetric)
    double4* rec = reinterpret_cast<double4*>(d.lmrec) + 4 * (size_t)lm;
    rec[0] = d.lms_lin4[lm];
    rec[1] = d.jl_scale4[lm];
    rec[2] = make_double4(Hi[0], Hi[1], Hi[2], Hi[4]);
    rec[3] = make_double4(Hi[5], Hi[8], 0, 0);
    if (d.v2.lmrec && !d.prep_aux_only) {  // record of the lane-per-landmark kernel, one copy per lane the landmark occupies
      const int lp = d.v2.lm_pos[lm], pos = lp & ((1 << 26) - 1), lanes = ((lp >> 26) & 63) + 1;
      const double4 X = d.lms_lin4[lm], s = d.jl_scale4[lm];
      const double rv[14] = {X.x, X.y, X.z, X.w, s.x, s.y, s.z, s.w, Hi[0], Hi[1], Hi[2], Hi[4], Hi[5], Hi[8]};
      for (int q = 0; q < lanes; ++q) {
        double* r2 = d.v2.lmrec + ((size_t)(pos >> 6) * 14) * WAVE + (pos & 63) + q;
#pragma unroll
        for (int m = 0; m < 14; ++m) r2[m * WAVE] = rv[m];
      }
    }
  }
};

constexpr int HOT_REC_H = 12;  // double2 per camera record of the step-2 kernels: z (12 doubles), P (12 doubles)

// K10': right_mul_e0_joint (linearization_power_varproj.hpp:408-453); input z = sigma * (N_c x_c)
struct OpE0H {
  static constexpr int NRED = 3, NSC = 0;
  static constexpr bool CHECK_DONE = true;
  struct Local { HomObs o; };
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local& L, double* red) const {
    L.o.load(d, slot, cam, lm, uv);
    const double4* zc = reinterpret_cast<const double4*>(d.z) + 3 * cam;
    const double4 zz[3] = {zc[0], zc[1], zc[2]};
    double t[2];
    hom_jp_x(L.o.h, L.o.X, L.o.sw, zz, t);
#pragma unroll
    for (int j = 0; j < 3; ++j) red[j] += L.o.jl3[j] * t[0] + L.o.jl3[3 + j] * t[1];
  }
  __device__ void phase2(const Dp& d, int slot, int, int lm, double2, Local& L, const double* tot, double*) const {
    const double* Hi = d.hll_inv + 9 * (size_t)lm;
    const double v0 = Hi[0] * tot[0] + Hi[1] * tot[1] + Hi[2] * tot[2];
    const double v1 = Hi[3] * tot[0] + Hi[4] * tot[1] + Hi[5] * tot[2];
    const double v2 = Hi[6] * tot[0] + Hi[7] * tot[1] + Hi[8] * tot[2];
    const double s0 = L.o.jl3[0] * v0 + L.o.jl3[1] * v1 + L.o.jl3[2] * v2;
    const double s1 = L.o.jl3[3] * v0 + L.o.jl3[4] * v1 + L.o.jl3[5] * v2;
    store_q(d, slot, hom_q(L.o.h, L.o.sw, s0, s1));
  }
  __device__ void finish_lm(const Dp&, int, const double*) const {}
};

// K12': back_substitute_joint (landmark_block.hpp:574-623); d.z = sigma * (N_c inc_c)
struct OpBackJoint {
  static constexpr int NRED = 9, NSC = 1;
  static constexpr bool CHECK_DONE = false;
  struct Local { HomObs o; double4 r; double jpi[2]; };
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local& L, double* red) const {
    L.o.load(d, slot, cam, lm, uv);
    L.r = d.rres[slot];
    const double4* zc = reinterpret_cast<const double4*>(d.z) + 3 * cam;
    const double4 zz[3] = {zc[0], zc[1], zc[2]};
    hom_jp_x(L.o.h, L.o.X, L.o.sw, zz, L.jpi);
    acc_h6(red, L.o.jl3);
    const double a0 = L.r.x + L.jpi[0], a1 = L.r.y + L.jpi[1];
#pragma unroll
    for (int j = 0; j < 3; ++j) red[6 + j] += L.o.jl3[j] * a0 + L.o.jl3[3 + j] * a1;
  }
  __device__ static void delta4(const Dp& d, const double4& X, const double* tot, double (&dp)[4]) {
    double Hi[9];
    hinv_damped(tot, d.lambda_lm, Hi);
    const double d0 = -(Hi[0] * tot[6] + Hi[1] * tot[7] + Hi[2] * tot[8]);
    const double d1 = -(Hi[3] * tot[6] + Hi[4] * tot[7] + Hi[5] * tot[8]);
    const double d2 = -(Hi[6] * tot[6] + Hi[7] * tot[7] + Hi[8] * tot[8]);
    double w[4], beta;
    house4(X, w, beta);
    const double wd = w[1] * d0 + w[2] * d1 + w[3] * d2;  // inc_proj = Proj * inc (landmark_block.hpp:615)
    dp[0] = -beta * w[0] * wd;
    dp[1] = d0 - beta * w[1] * wd;
    dp[2] = d1 - beta * w[2] * wd;
    dp[3] = d2 - beta * w[3] * wd;
  }
  __device__ void phase2(const Dp& d, int, int, int, double2, Local& L, const double* tot, double* sc) const {
    double dp[4];
    delta4(d, L.o.X, tot, dp);
    const double rr[2] = {L.r.x, L.r.y};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double ji = L.jpi[r] + (L.o.jl4[4 * r] * dp[0] + L.o.jl4[4 * r + 1] * dp[1] + L.o.jl4[4 * r + 2] * dp[2] +
                                    L.o.jl4[4 * r + 3] * dp[3]);
      sc[0] -= ji * (0.5 * ji + rr[r]);
    }
  }
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    double dp[4];
    delta4(d, d.lms_lin4[lm], tot, dp);
    const double4 s = d.jl_scale4[lm];
    double4 X = d.lms4[lm];
    X.x += dp[0] * s.x; X.y += dp[1] * s.y; X.z += dp[2] * s.z; X.w += dp[3] * s.w;  // :621-622
    d.lms4[lm] = X;
  }
};

// Per-term landmark-major kernel of step 2 with the LDS camera cache + LDS accumulation of the hot
// cameras: the step-2 twin of e0_lm_cached<true> (same pipeline, 2-row tiles, camera record = z_c (12)
// + full P_c (12) = 192 B, landmark record = X | s | Hll^-1 = 128 B).

POVAR_KERNEL __launch_bounds__(E0C_BLOCK) void e0_lm_cached_h(Dp d, int bins_per_wg, double* hot_out) {
  if (d.flags[1]) return;
  extern __shared__ double2 hot[];  // [n_hot][HOT_REC_H] records, then acc[12][n_hot]
  const int n_hot = d.n_hot_acc;
  double* acc = reinterpret_cast<double*>(hot + n_hot * HOT_REC_H);
  for (int i = threadIdx.x; i < n_hot * 12; i += E0C_BLOCK) acc[i] = 0;
  {
    const double2* src = reinterpret_cast<const double2*>(d.hot_rec);  // [rank][12 double2]: z, P
    for (int i = threadIdx.x; i < n_hot * HOT_REC_H; i += E0C_BLOCK) hot[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bin0 = blockIdx.x * bins_per_wg;
  const int bin1 = min(bin0 + bins_per_wg, d.n_bins);
  constexpr int STRIDE = E0C_BLOCK / WAVE;
  int n_meta = lane | (lane << 8), n_cam = 0, n_lm = 0, n_cpos = 0;
  double2 n_uv = make_double2(0, 0);
  if (bin0 + wave < bin1) {
    const int s = (bin0 + wave) * WAVE + lane;
    n_meta = d.meta[s]; n_cam = d.cam[s]; n_lm = d.lm[s]; n_uv = d.uv[s]; n_cpos = d.cold_pos[s];
  }
  for (int bin = bin0 + wave; bin < bin1; bin += STRIDE) {
    const int slot = bin * WAVE + lane;
    const int meta = n_meta, cam = n_cam, lm = n_lm, cpos = n_cpos;
    const double2 uv = n_uv;
    const bool valid = (meta & META_REAL) && !(meta & META_LONG);
    const int seg_first = meta & 255, seg_last = (meta >> 8) & 255;
    const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 4 * (size_t)(valid ? lm : 0);
    const double4 X = rec[0], s4 = rec[1], h0 = rec[2], h1 = rec[3];
    if (bin + STRIDE < bin1) {
      const int s = slot + STRIDE * WAVE;
      n_meta = d.meta[s]; n_cam = d.cam[s]; n_lm = d.lm[s]; n_uv = d.uv[s]; n_cpos = d.cold_pos[s];
    }
    double red[3] = {0, 0, 0};
    double jl3[6];
    Hom h;
    double sw = 1.0;
    const int hr = ((meta >> META_HOT_SHIFT) & META_HOT_MASK);
    const bool is_hot = hr > 0 && hr <= n_hot;
    if (valid) {
      Cam P;
      double4 zz[3];
      if (is_hot) {
        const double2* hp = hot + (hr - 1) * HOT_REC_H;
        const double2 a0 = hp[0], a1 = hp[1], a2 = hp[2], a3 = hp[3], a4 = hp[4], a5 = hp[5];
        zz[0] = make_double4(a0.x, a0.y, a1.x, a1.y);
        zz[1] = make_double4(a2.x, a2.y, a3.x, a3.y);
        zz[2] = make_double4(a4.x, a4.y, a5.x, a5.y);
        const double2 b0 = hp[6], b1 = hp[7], b2 = hp[8], b3 = hp[9], b4 = hp[10], b5 = hp[11];
        P.r0 = make_double4(b0.x, b0.y, b1.x, b1.y);
        P.r1 = make_double4(b2.x, b2.y, b3.x, b3.y);
        P.r2 = make_double4(b4.x, b4.y, b5.x, b5.y);
      } else {
        P = load_cam(d.cams_lin4, cam);
        const double4* zc = reinterpret_cast<const double4*>(d.z) + 3 * cam;
        zz[0] = zc[0]; zz[1] = zc[1]; zz[2] = zc[2];
      }
      sw = d.robust ? d.sw[slot] : 1.0;
      h = hom_project(P, X, uv.x, uv.y);
      double jl4[8], w[4], beta;
      hom_jl4(P, h, sw, s4, jl4);
      house4(X, w, beta);
      jl3_of_jl4(jl4, w, beta, jl3);
      double t[2];
      hom_jp_x(h, X, sw, zz, t);
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) red[jj] += jl3[jj] * t[0] + jl3[3 + jj] * t[1];
    }
    seg_reduce_steps<3>(red, lane, seg_first, seg_last,
                        __builtin_amdgcn_readfirstlane((meta >> META_STEPS_SHIFT) & 7));
    if (valid) {
      const double v0 = h0.x * red[0] + h0.y * red[1] + h0.z * red[2];
      const double v1 = h0.y * red[0] + h0.w * red[1] + h1.x * red[2];
      const double v2 = h0.z * red[0] + h1.x * red[1] + h1.y * red[2];
      const double s0 = jl3[0] * v0 + jl3[1] * v1 + jl3[2] * v2;
      const double s1 = jl3[3] * v0 + jl3[4] * v1 + jl3[5] * v2;
      const double4 q = hom_q(h, sw, s0, s1);
      if (is_hot) {
        double* a = acc + (hr - 1);
        const double v[12] = {X.x * q.x, X.y * q.x, X.z * q.x, X.w * q.x, X.x * q.y, X.y * q.y,
                              X.z * q.y, X.w * q.y, X.x * q.z, X.y * q.z, X.z * q.z, X.w * q.z};
#pragma unroll
        for (int jj = 0; jj < 12; ++jj)
          __hip_atomic_fetch_add(a + jj * n_hot, v[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        d.q4c[cpos] = q;  // cold camera: straight to its place in the cold camera-major view
      }
    }
  }
  if (d.long_in_kernel) {
    // landmarks with more than 64 observations: one wavefront per landmark, two passes over its slots with the
    // LDS camera cache and accumulators (see e0_lm_cached<true>)
    for (int j = blockIdx.x * STRIDE + wave; j < d.n_long; j += gridDim.x * STRIDE) {
      const int lm = d.long_lm[j], first = d.long_first[j], cnt = d.long_cnt[j];
      const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 4 * (size_t)lm;
      const double4 X = rec[0], s4 = rec[1], h0 = rec[2], h1 = rec[3];
      double hw[4], hbeta;
      house4(X, hw, hbeta);
      double tot[3] = {0, 0, 0};
      for (int pass = 0; pass < 2; ++pass) {
        for (int i0 = 0; i0 < cnt; i0 += WAVE) {
          const bool in = i0 + lane < cnt;
          const int slot = first + (in ? i0 + lane : 0);
          double red[3] = {0, 0, 0};
          double jl3[6];
          Hom h;
          double sw = 1.0;
          int hr = 0;
          if (in) {
            const int meta = d.meta[slot];
            hr = (meta >> META_HOT_SHIFT) & META_HOT_MASK;
            const double2 uv = d.uv[slot];
            Cam P;
            double4 zz[3];
            if (hr > 0 && hr <= n_hot) {
              const double2* hp = hot + (hr - 1) * HOT_REC_H;
              const double2 a0 = hp[0], a1 = hp[1], a2 = hp[2], a3 = hp[3], a4 = hp[4], a5 = hp[5];
              zz[0] = make_double4(a0.x, a0.y, a1.x, a1.y);
              zz[1] = make_double4(a2.x, a2.y, a3.x, a3.y);
              zz[2] = make_double4(a4.x, a4.y, a5.x, a5.y);
              const double2 b0 = hp[6], b1 = hp[7], b2 = hp[8], b3 = hp[9], b4 = hp[10], b5 = hp[11];
              P.r0 = make_double4(b0.x, b0.y, b1.x, b1.y);
              P.r1 = make_double4(b2.x, b2.y, b3.x, b3.y);
              P.r2 = make_double4(b4.x, b4.y, b5.x, b5.y);
            } else {
              const int cam = d.cam[slot];
              P = load_cam(d.cams_lin4, cam);
              const double4* zc = reinterpret_cast<const double4*>(d.z) + 3 * cam;
              zz[0] = zc[0]; zz[1] = zc[1]; zz[2] = zc[2];
            }
            sw = d.robust ? d.sw[slot] : 1.0;
            h = hom_project(P, X, uv.x, uv.y);
            double jl4[8];
            hom_jl4(P, h, sw, s4, jl4);
            jl3_of_jl4(jl4, hw, hbeta, jl3);
            double t[2];
            hom_jp_x(h, X, sw, zz, t);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) red[jj] += jl3[jj] * t[0] + jl3[3 + jj] * t[1];
          }
          if (pass == 0) {
            wave_sum<3>(red);
            tot[0] += red[0]; tot[1] += red[1]; tot[2] += red[2];
          } else if (in) {
            const double v0 = h0.x * tot[0] + h0.y * tot[1] + h0.z * tot[2];
            const double v1 = h0.y * tot[0] + h0.w * tot[1] + h1.x * tot[2];
            const double v2 = h0.z * tot[0] + h1.x * tot[1] + h1.y * tot[2];
            const double s0 = jl3[0] * v0 + jl3[1] * v1 + jl3[2] * v2;
            const double s1 = jl3[3] * v0 + jl3[4] * v1 + jl3[5] * v2;
            const double4 q = hom_q(h, sw, s0, s1);
            if (hr > 0 && hr <= n_hot) {
              double* a = acc + (hr - 1);
              const double v[12] = {X.x * q.x, X.y * q.x, X.z * q.x, X.w * q.x, X.x * q.y, X.y * q.y,
                                    X.z * q.y, X.w * q.y, X.x * q.z, X.y * q.z, X.z * q.z, X.w * q.z};
#pragma unroll
              for (int jj = 0; jj < 12; ++jj)
                __hip_atomic_fetch_add(a + jj * n_hot, v[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              d.q4c[d.cold_pos[slot]] = q;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_hot * 12; i += E0C_BLOCK)
    hot_out[((size_t)(i / 12) * gridDim.x + blockIdx.x) * 12 + i % 12] = acc[(i % 12) * n_hot + i / 12];
}

// Gram moments of the unscaled weighted Jp12: Jp12^T Jp12 = w * (C (x) X X^T),
// C = [[D00^2, 0, D00 D02], [0, D00^2, D00 D12], [., ., D02^2 + D12^2]]
POVAR_KERNEL __launch_bounds__(256) void cm_gram_h(Dp d, int gather) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (item >= d.n_items) return;
  const int b = d.item_off[item], e = d.item_off[item + 1];
  const Cam P = load_cam(d.cams_lin4, d.item_cam[item]);
  double acc[40];
#pragma unroll
  for (int k = 0; k < 40; ++k) acc[k] = 0;
  for (int p = b + lane; p < e; p += WAVE) {
    double sw = 1.0;
    if (d.robust && !gather) sw = d.sw[d.cm_slot[p]];
    const double4 X = gather ? d.lms_lin4[d.cm_lm[p]]
                             : make_double4(d.cm_h[p], d.cm_h[d.n_obs + p], d.cm_h[2 * d.n_obs + p], d.cm_h[3 * d.n_obs + p]);
    const double2 uv = d.cm_uv[p];
    const Hom h = hom_project(P, X, uv.x, uv.y);
    if (d.robust && gather) {  // the per-slot sqrt(w) is not kept in the lane-per-landmark mode: recompute it
      double e_, w_;
      error_weight(d, h.r0 * h.r0 + h.r1 * h.r1, e_, w_);
      sw = sqrt(w_);
    }
    const double w = sw * sw;
    const double m[4] = {w * h.D00 * h.D00, w * h.D00 * h.D02, w * h.D00 * h.D12, w * (h.D02 * h.D02 + h.D12 * h.D12)};
    const double hh[10] = {X.x * X.x, X.x * X.y, X.x * X.z, X.x * X.w, X.y * X.y,
                           X.y * X.z, X.y * X.w, X.z * X.z, X.z * X.w, X.w * X.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[10 * k + j] += m[k] * hh[j];
  }
  wave_sum<40>(acc);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 40; ++k) d.item_partG[40 * (size_t)item + k] = acc[k];
  }
}

// per camera after the Gram sums: diag2 / sigma (get_Jp_diag2_projective_space,
// linearization_varproj.hpp:225-264; linearizor_power_varproj.cpp:97-105) and the Householder
// vector of vec(P_c) for the tangent basis N_c
POVAR_KERNEL __launch_bounds__(CFL_THREADS) void cam_finish_linearize_h(Dp d, const double* G_in, double* ncw) {
  const int c = blockIdx.x;
  constexpr int NQ = CFL_THREADS / 64;
  __shared__ double part[NQ][40];
  __shared__ double g[40];
  if (G_in) {
    if (threadIdx.x < 40) g[threadIdx.x] = G_in[40 * (size_t)c + threadIdx.x];
  } else {
    const int e = threadIdx.x % 64, q = threadIdx.x / 64;
    if (e < 40) {
      double s = 0;
      for (int it = d.cam_item_off[c] + q; it < d.cam_item_off[c + 1]; it += NQ) s += d.item_partG[40 * (size_t)it + e];
      part[q][e] = s;
    }
    __syncthreads();
    if (threadIdx.x < 40) {
      double sum = 0;  // fixed order
#pragma unroll
      for (int k = 0; k < NQ; ++k) sum += part[k][threadIdx.x];
      g[threadIdx.x] = sum;
    }
  }
  __syncthreads();
  if (threadIdx.x < 40) d.G[40 * (size_t)c + threadIdx.x] = g[threadIdx.x];
  if (threadIdx.x < 12) {
    const int blk = threadIdx.x >> 2, j = threadIdx.x & 3;
    const int dj = sym10(j, j);
    const double v = blk < 2 ? g[dj] : g[30 + dj];
    d.diag2[12 * (size_t)c + threadIdx.x] = v;
    d.sigma[12 * (size_t)c + threadIdx.x] = 1.0 / (d.eps + sqrt(v));
  }
  if (threadIdx.x == 0) {
    const double* P = reinterpret_cast<const double*>(d.cams_lin4) + 12 * (size_t)c;
    double nv = 0;
    for (int k = 0; k < 12; ++k) nv += P[k] * P[k];
    nv = sqrt(nv);
    double w[12], wtw = 0;
    for (int k = 0; k < 12; ++k) w[k] = P[k];
    w[0] += P[0] >= 0 ? nv : -nv;
    for (int k = 0; k < 12; ++k) wtw += w[k] * w[k];
    for (int k = 0; k < 12; ++k) ncw[13 * (size_t)c + k] = w[k];
    ncw[13 * (size_t)c + 12] = 2.0 / wtw;
  }
}

// K8': B_c = N_c^T (Hpp12 + lambda I) N_c = N_c^T Hpp12 N_c + lambda I_11, Cholesky inverse 11x11
// (linearization_power_varproj.hpp:91-121).  Sixteen lanes per camera as cam_build_binv: lane i owns row i of the two
// projections, then chol_inverse_16<11>.
POVAR_KERNEL __launch_bounds__(K8_THREADS) void cam_build_binv_h(Dp d, double lambda, const double* ncw) {
  __shared__ double As[K8_CAMS_PER_WG][144];
  __shared__ double Ts[K8_CAMS_PER_WG][144];
  const int q = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int c = blockIdx.x * K8_CAMS_PER_WG + q;
  const bool in = c < d.n_cams;
  double* A = As[q];
  double* T = Ts[q];
  const double* w = ncw + 13 * (size_t)(in ? c : 0);
  const double beta = w[12];
  if (in) {
    const double* g = d.G + 40 * (size_t)c;
    const double* sg = d.sigma + 12 * (size_t)c;
    for (int e = l; e < 144; e += 16) {
      const int r = e / 12, cc = e % 12, a = r >> 2, i = r & 3, b = cc >> 2, j = cc & 3;
      const int ij = sym10(i, j);
      double v;
      if (a == b) v = a < 2 ? g[ij] : g[30 + ij];
      else if (a + b == 1) v = 0;
      else v = g[10 * ((a == 2 ? b : a) + 1) + ij];
      A[e] = v * sg[r] * sg[cc];
    }
  }
  __syncthreads();
  // T = A N (12 x 11): row i by lane i
  if (in && l < 12) {
    double aw = 0;
    for (int k = 0; k < 12; ++k) aw += A[l * 12 + k] * w[k];
    for (int j = 0; j < 11; ++j) T[l * 12 + j] = A[l * 12 + j + 1] - beta * aw * w[j + 1];
  }
  __syncthreads();
  // M = N^T T (11 x 11) back into A (row stride kept at 12): column j by lane j
  if (in && l < 11) {
    double wt = 0;
    for (int k = 0; k < 12; ++k) wt += w[k] * T[k * 12 + l];
    for (int i = 0; i < 11; ++i) A[i * 12 + l] = T[(i + 1) * 12 + l] - beta * w[i + 1] * wt + (i == l ? lambda : 0.0);
  }
  if (!in)
    for (int e = l; e < 144; e += 16) A[e] = (e / 12 == e % 12) ? 1.0 : 0.0;
  __syncthreads();
  chol_inverse_16<11>(A, l, in ? d.binv + 144 * (size_t)c : nullptr);  // 11x11 row-major in the first 121 entries
}

// tangent projection of a per-camera 12-vector: out11 = N_c^T (in12)
__device__ inline void nt_apply(const double* w, double beta, const double (&y)[12], double (&o)[11]) {
  double wy = 0;
#pragma unroll
  for (int k = 0; k < 12; ++k) wy += w[k] * y[k];
#pragma unroll
  for (int j = 0; j < 11; ++j) o[j] = y[j + 1] - beta * w[j + 1] * wy;
}

// b11_c = N_c^T (sigma * sum_items); one wavefront per camera
POVAR_KERNEL __launch_bounds__(256) void cam_sum_items_h(Dp d, double* out11, const double* ncw) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d.n_cams) return;
  double y[12];
  camera_item_sum(d, c, lane, y);
#pragma unroll
  for (int j = 0; j < 12; ++j) y[j] *= d.sigma[12 * (size_t)c + j];
  double o[11];
  nt_apply(ncw + 13 * (size_t)c, ncw[13 * (size_t)c + 12], y, o);
  if (lane < 11) {
    double v = 0;
#pragma unroll
    for (int j = 0; j < 11; ++j) v = (lane == j) ? o[j] : v;
    out11[11 * (size_t)c + lane] = v;
  }
}

// K9' + K11': tmp11 = B^-1 y11, accum11 (+)= tmp11, z = sigma * (N_c tmp11)
// (right_mul_b_inv_joint + loop body of solve_joint, linearization_power_varproj.hpp:246-257, 342-360).
// mode 0: y11 = -b11; 1: y12 = sigma * sum of scatter items, y11 = N^T y12; 2: y12 = dense d.y (all-reduced)
POVAR_KERNEL __launch_bounds__(K9_CAMS * 64) void cam_binv_axpy_h(Dp d, int mode, int want_norms, const double* ncw) {
  if (mode != 0 && d.flags[1]) return;
  __shared__ double sh[K9_CAMS * 2];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * K9_CAMS + (threadIdx.x >> 6);
  const bool in = c < d.n_cams;
  double y11[11];
#pragma unroll
  for (int j = 0; j < 11; ++j) y11[j] = 0;
  const double* w = ncw + 13 * (size_t)(in ? c : 0);
  const double beta = w[12];
  if (in) {
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < 11; ++j) y11[j] = -d.b[11 * (size_t)c + j];
    } else {
      double y[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) y[j] = 0;
      if (mode == 1) {
        camera_item_sum(d, c, lane, y);
#pragma unroll
        for (int j = 0; j < 12; ++j) y[j] *= d.sigma[12 * (size_t)c + j];
      } else {
#pragma unroll
        for (int j = 0; j < 12; ++j) y[j] = d.y[12 * (size_t)c + j];
      }
      nt_apply(w, beta, y, y11);
    }
  }
  double nrm[2] = {0, 0};
  double s = 0;
  if (in && lane < 11) {
    const double* Bi = d.binv + 144 * (size_t)c + 11 * lane;
#pragma unroll
    for (int j = 0; j < 11; ++j) s += Bi[j] * y11[j];
    const size_t idx = 11 * (size_t)c + lane;
    const double acc = mode == 0 ? s : d.accum[idx] + s;
    d.tmp[idx] = s;
    d.accum[idx] = acc;
    nrm[0] = s * s;
    nrm[1] = acc * acc;
  }
  // p12 = N_c tmp11: p_i = [0; tmp]_i - beta w_i (w[1:] . tmp); lane i < 12 needs tmp_{i-1} and the dot
  double wt = (in && lane < 11) ? w[lane + 1] * s : 0.0;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) wt += shfl_xor_d(wt, m);  // lanes 0..15 hold the 11 products
  const double prev = shfl_up_d(s, 1);
  if (in && lane < 12) {
    const double p = (lane == 0 ? 0.0 : prev) - beta * w[lane] * wt;
    store_z(d, c, lane, p * d.sigma[12 * (size_t)c + lane]);
    if (mode == 2) d.y[12 * (size_t)c + lane] = 0;
  }
  if (want_norms) {
    block_sum<2, K9_CAMS * 64>(nrm, sh);
    if (threadIdx.x == 0) {
      d.norm_part[2 * (size_t)blockIdx.x] = nrm[0];
      d.norm_part[2 * (size_t)blockIdx.x + 1] = nrm[1];
    }
  }
}

// K10' on the lane-per-landmark layout (right_mul_e0_joint, linearization_power_varproj.hpp:408-453): e0_lpl's
// structure (povar_kernels.hpp) with the 2-row tiles of step 2.  Records are 192 bytes (z = sigma * N_c x_c, then the
// full P_c), the landmark lane carries X, the Jl column scale and Hll^-1 (the column scale cannot be folded into
// Hll^-1 here: the tangent basis N_l sits between them).
constexpr int LPL_REC_H = 14;  // doubles per landmark lane in V2::lmrec (step 2)
// LDS stride of a camera record in the lane-per-landmark kernels of step 2, in double2: the record has 12 (192 B), but
// a 12-quad stride puts every record on one of FOUR bank-quad classes (12 s mod 16) -- four-way ds_read_b128 conflicts
// whatever the placement does; 13 gives the slot -> class bijection of step 1 (11 s mod 16), which the row placement of
// the layout (lpl_layout.hpp) is built for
#ifndef POVAR_LPL_CAMREC_H
#define POVAR_LPL_CAMREC_H 13
#endif
constexpr int LPL_CAMREC_H = POVAR_LPL_CAMREC_H;
__host__ __device__ inline size_t lpl_lds_bytes_h(int n_hot) {
  return (size_t)n_hot * LPL_CAMREC_H * sizeof(double2) + (size_t)(n_hot + 3 * lpl_hubs(n_hot)) * 96 + 16;
}
template <bool ROBUST>
__global__ __launch_bounds__(E0C_BLOCK) void e0_lpl_h(Dp d, double* hot_out) {
  const int done = d.flags[1];  // requested first, tested after the LDS staging (no global side effects before)
  extern __shared__ double2 hot[];  // [n_hot][HOT_REC_H] records, then acc[12][n_slots], then the tile counter
  const V2& v = d.v2;
  // this workgroup's camera slots: the records of the cameras it keeps in LDS (lpl_layout.hpp) and their accumulators
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_hot = v.wg_cam_off[blockIdx.x + 1] - cam0;
  const int hubs = v.hubs, n_slots = n_hot + 3 * hubs;
  double* acc = reinterpret_cast<double*>(hot + n_hot * LPL_CAMREC_H);
  int* grab_ctr = reinterpret_cast<int*>(acc + n_slots * 12);
  for (int i = threadIdx.x; i < n_slots * 12; i += E0C_BLOCK) acc[i] = 0;
  if (threadIdx.x == 0) *grab_ctr = 0;
  const double2* rec_img = reinterpret_cast<const double2*>(d.hot_rec);
  {
    // staging: slot -> camera rank -> record, two dependent L2 round trips; every thread first requests all its
    // ranks, then all its record pieces, then writes LDS (a plain loop pays the two latencies once per pass)
    constexpr int PASSES = (HOT_ACC_MAX * HOT_REC_H + E0C_BLOCK - 1) / E0C_BLOCK;
    int rk[PASSES];
    double2 piece[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int i = threadIdx.x + u * E0C_BLOCK;
      rk[u] = i < n_hot * HOT_REC_H ? v.wg_cams[cam0 + i / HOT_REC_H] : 0;
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int i = threadIdx.x + u * E0C_BLOCK;
      piece[u] = rec_img[(size_t)rk[u] * (HOT_REC_STRIDE / 2) + i % HOT_REC_H];
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int i = threadIdx.x + u * E0C_BLOCK;
      if (i < n_hot * HOT_REC_H) hot[(i / HOT_REC_H) * LPL_CAMREC_H + i % HOT_REC_H] = piece[u];
    }
  }
  __syncthreads();
  if (done) return;
  const int lane = threadIdx.x & 63;
  // The workgroup's tiles are sorted longest first; its wavefronts take them on demand (one LDS counter), so a
  // wavefront's last tile is a short one.  The workgroups carry equal observation totals (lpl_layout.hpp).
  const int t_begin = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t_end = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  auto grab = [&]() -> int {
    int n = 0;
    if (lane == 0) n = __hip_atomic_fetch_add(grab_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    n = __builtin_amdgcn_readfirstlane(n);
    const long long t = (long long)t_begin + n;
    return t < t_end ? (int)t : t_end;
  };

  // tile table through the scalar cache (constant address space + wave-uniform index => s_load_dwordx4): a vector
  // load here would put a vmcnt(0) drain inside the row pipeline
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)v.tile;
  auto tile_info = [&](int t, int& row0, int& k, int& nh, int& fl) {
    row0 = tiles[4 * t];
    k = tiles[4 * t + 1];
    nh = tiles[4 * t + 2];
    fl = tiles[4 * t + 3];
  };
  LplCursor pc;
  pc.t = grab();
  pc.pass = 0;
  pc.j = 0;
  pc.row0 = 0;
  pc.k = 1;
  int c_t = pc.t, c_row0 = 0, c_k = 0, c_nh = 0, c_fl = 0;
  // the tile after the one being consumed, taken when the consumer enters a tile: the prefetch cursor runs at most
  // LPL_DEPTH = 3 rows ahead and a tile has at least 4 row steps, so it never needs more than this one
  int nx_t = t_end;
  if (c_t < t_end) {
    tile_info(c_t, c_row0, c_k, c_nh, c_fl);
    pc.row0 = c_row0;
    pc.k = c_k;
    nx_t = grab();
  }
  // request the row under the prefetch cursor and advance it
  auto issue = [&](LplRow& r) {
    if (pc.t < t_end) {
      // the backward pass walks the rows in reverse: the rows read last are the ones most likely still in L2
      const size_t i = ((size_t)pc.row0 + (pc.pass ? pc.k - 1 - pc.j : pc.j)) * WAVE + lane;
      r.uv = v.uv[i];
      r.cw = v.cw[i];
      if (ROBUST) r.w = v.w[i];
      if (++pc.j == pc.k) {
        pc.j = 0;
        if (++pc.pass == 2) {
          pc.pass = 0;
          pc.t = nx_t;
          if (pc.t < t_end) {
            int nh_, fl_;
            tile_info(pc.t, pc.row0, pc.k, nh_, fl_);
          }
        }
      }
    }
  };
  LplRow n1, n2, n3;
  n1.cw = n2.cw = n3.cw = -1;
  n1.w = n2.w = n3.w = 1.0;
  n1.uv = n2.uv = n3.uv = make_double2(0, 0);
  issue(n1);
  issue(n2);
  issue(n3);
  // landmark record of the lane (step 2): X (4), Jl column scale s (4), Hll^-1 (6) = 14 entries of lmrec[tile][14][64];
  // the Householder vector of X (tangent basis N_l, landmark_block.hpp:227-269) is rebuilt once per tile
  auto load_rec = [&](int t, double4& X, double4& s4, double (&Hi)[6]) {
    const double* rp = v.lmrec + ((size_t)t * LPL_REC_H) * WAVE + lane;
    X = make_double4(rp[0], rp[WAVE], rp[2 * WAVE], rp[3 * WAVE]);
    s4 = make_double4(rp[4 * WAVE], rp[5 * WAVE], rp[6 * WAVE], rp[7 * WAVE]);
#pragma unroll
    for (int m = 0; m < 6; ++m) Hi[m] = rp[(8 + m) * WAVE];
  };
  auto read_cam = [&](const double2* hp, Cam& P) {  // entries 12..23 of a record: P row-major
    const double2 b0 = hp[6], b1 = hp[7], b2 = hp[8], b3 = hp[9], b4 = hp[10], b5 = hp[11];
    P.r0 = make_double4(b0.x, b0.y, b1.x, b1.y);
    P.r1 = make_double4(b2.x, b2.y, b3.x, b3.y);
    P.r2 = make_double4(b4.x, b4.y, b5.x, b5.y);
  };
  auto read_z = [&](const double2* hp, double4 (&zz)[3]) {
    const double2 a0 = hp[0], a1 = hp[1], a2 = hp[2], a3 = hp[3], a4 = hp[4], a5 = hp[5];
    zz[0] = make_double4(a0.x, a0.y, a1.x, a1.y);
    zz[1] = make_double4(a2.x, a2.y, a3.x, a3.y);
    zz[2] = make_double4(a4.x, a4.y, a5.x, a5.y);
  };
  double4 X = make_double4(0, 0, 0, 1), s4 = make_double4(1, 1, 1, 1);
  double Hi[6] = {0, 0, 0, 0, 0, 0};
  if (c_t < t_end) load_rec(c_t, X, s4, Hi);
  while (c_t < t_end) {
    double hw[4], hbeta;
    house4(X, hw, hbeta);
    double red[3] = {0, 0, 0};
    for (int j = 0; j < c_k; ++j) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      // The record is read through an LDS pointer or a global pointer, never through a select of the two: a generic
      // pointer makes every read a flat_load (texture path into the LDS, both wait counters).  Rows [0, c_nh) have an
      // LDS-resident camera in every lane: wave-uniform branch, LDS reads only.
      auto fwd = [&](const Cam& P, const double4 (&zz)[3]) {
        const double sw = ROBUST ? sqrt(cur.w) : 1.0;
        const Hom h = hom_project(P, X, cur.uv.x, cur.uv.y);
        double jl4[8], jl3[6], t[2];
        hom_jl4(P, h, sw, s4, jl4);
        jl3_of_jl4(jl4, hw, hbeta, jl3);
        hom_jp_x(h, X, sw, zz, t);
#pragma unroll
        for (int m = 0; m < 3; ++m) red[m] += jl3[m] * t[0] + jl3[3 + m] * t[1];
      };
      Cam P;
      double4 zz[3];
      if (j < c_nh) {
        const double2* hp = hot + lpl_cw_slot(cur.cw) * LPL_CAMREC_H;
        read_z(hp, zz);
        read_cam(hp, P);
        fwd(P, zz);
      } else if (cur.cw != -1) {
        if (cur.cw >= 0) {
          const double2* hp = hot + lpl_cw_slot(cur.cw) * LPL_CAMREC_H;
          read_z(hp, zz);
          read_cam(hp, P);
        } else {
          const double2* hp = rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2);
          read_z(hp, zz);
          read_cam(hp, P);
        }
        fwd(P, zz);
      }
    }
    if (c_fl & 1) {  // landmarks dealt over several lanes: sum their partial u = Jl^T t (segmented wavefront scan)
      const int sg = v.seg[(size_t)c_t * WAVE + lane];
      seg_reduce_steps<3>(red, lane, sg & 255, (sg >> 8) & 255, 4);
    }
    const double g[3] = {Hi[0] * red[0] + Hi[1] * red[1] + Hi[2] * red[2], Hi[1] * red[0] + Hi[3] * red[1] + Hi[4] * red[2],
                         Hi[2] * red[0] + Hi[4] * red[1] + Hi[5] * red[2]};
    // the next tile's record (Hll^-1 is dead by now; X and s are still needed: second set)
    const int n_t = nx_t;
    double4 nX = make_double4(0, 0, 0, 1), ns4 = make_double4(1, 1, 1, 1);
    double nHi[6] = {0, 0, 0, 0, 0, 0};
    if (n_t < t_end) load_rec(n_t, nX, ns4, nHi);
    for (int jj = 0; jj < c_k; ++jj) {
      const int j = c_k - 1 - jj;
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      auto bwd = [&](const Cam& P) -> double4 {
        const double sw = ROBUST ? sqrt(cur.w) : 1.0;
        const Hom h = hom_project(P, X, cur.uv.x, cur.uv.y);
        double jl4[8], jl3[6];
        hom_jl4(P, h, sw, s4, jl4);
        jl3_of_jl4(jl4, hw, hbeta, jl3);
        const double s0 = jl3[0] * g[0] + jl3[1] * g[1] + jl3[2] * g[2];
        const double s1 = jl3[3] * g[0] + jl3[4] * g[1] + jl3[5] * g[2];
        return hom_q(h, sw, s0, s1);
      };
      auto scatter = [&](const double4& q) {
        double* a = acc + lpl_acc_slot(cur.cw, hubs);  // acc[m][slot]: consecutive slots on consecutive banks
        const double val[12] = {X.x * q.x, X.y * q.x, X.z * q.x, X.w * q.x, X.x * q.y, X.y * q.y,
                                X.z * q.y, X.w * q.y, X.x * q.z, X.y * q.z, X.z * q.z, X.w * q.z};
#pragma unroll
        for (int m = 0; m < 12; ++m)
          __hip_atomic_fetch_add(a + m * n_slots, val[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      Cam P;
      if (j < c_nh) {  // wave-uniform: LDS only
        read_cam(hot + lpl_cw_slot(cur.cw) * LPL_CAMREC_H, P);
        scatter(bwd(P));
      } else if (cur.cw >= 0) {
        read_cam(hot + lpl_cw_slot(cur.cw) * LPL_CAMREC_H, P);
        scatter(bwd(P));
      } else if (cur.cw < -1) {
        // where q goes: row-major next to the other lanes' (graphs with many cold observations: the per-camera kernel
        // gathers, Dp::q_rows) or straight to its place in the camera-major cold view (few: one 32-byte store per lane)
        const int cold_at = d.q_rows ? lpl_cold_q(c_fl, c_nh, j, lane) : v.cpos[((size_t)c_row0 + j) * WAVE + lane];
        read_cam(rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2), P);
        d.q4c[cold_at] = bwd(P);
      }
    }
    c_t = n_t;
    if (c_t < t_end) {
      tile_info(c_t, c_row0, c_k, c_nh, c_fl);
      nx_t = grab();
    }
    X = nX;
    s4 = ns4;
#pragma unroll
    for (int m = 0; m < 6; ++m) Hi[m] = nHi[m];
  }
  __syncthreads();
  // accumulators -> this workgroup's partial records (camera-major in hot_out: the per-camera kernel reads one run);
  // 16-byte stores, all record indices requested first (one L2 round trip, not one per pass)
  {
    constexpr int PASSES = (HOT_ACC_MAX * 6 + E0C_BLOCK - 1) / E0C_BLOCK;
    int rec[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int i = threadIdx.x + u * E0C_BLOCK;
      rec[u] = i < n_hot * 6 ? v.wg_slot_rec[cam0 + i / 6] : 0;
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int i = threadIdx.x + u * E0C_BLOCK;
      if (i < n_hot * 6) {
        const int r = i / 6, m = 2 * (i % 6);
        const double* a0 = acc + m * n_slots;
        const double* a1 = a0 + n_slots;
        double2 s;
        if (r < hubs) {
          s.x = (a0[4 * r] + a0[4 * r + 1]) + (a0[4 * r + 2] + a0[4 * r + 3]);
          s.y = (a1[4 * r] + a1[4 * r + 1]) + (a1[4 * r + 2] + a1[4 * r + 3]);
        } else {
          s.x = a0[r + 3 * hubs];
          s.y = a1[r + 3 * hubs];
        }
        reinterpret_cast<double2*>(hot_out + (size_t)rec[u] * 12)[i % 6] = s;
      }
    }
  }
  if (d.p2p_epoch && blockIdx.x == 0 && threadIdx.x == 0) *d.p2p_epoch += 1;  // one tick per term, read by the next kernels
}


// K7' on the lane-per-landmark layout: get_Hll_inv_add_Hpp_b_joint (landmark_block.hpp:474-507), the landmark half
// of prepare_Hb_joint -- prepare_lpl (povar_kernels.hpp) with the homogeneous tile.  Forward pass: Hll = Jl3^T Jl3
// and Jl3^T r in registers; the lane inverts Hll + lambda I, stores Hll^-1 and the landmark records of the per-term
// kernels; backward pass: Jp^T (r - Jl3 w) into the camera accumulators (12 ambient values per observation; the
// tangent projection N_c^T is applied per camera afterwards, cam_nt_project).  Replaces lm_regular<OpPrepareH> +
// cm_scatter + cam_sum_items_h (495 + 124 + 6 us on venice-1778) for the LDSACC mode.
template <bool ROBUST>
__global__ __launch_bounds__(E0C_BLOCK) void prepare_lpl_h(Dp d, double* hot_out) {
  extern __shared__ double2 hot[];  // [n_hot][PREP_REC] records (P_c row-major), then acc[12][n_slots], then the tile counter
  const V2& v = d.v2;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_hot = v.wg_cam_off[blockIdx.x + 1] - cam0;
  const int hubs = v.hubs, n_slots = n_hot + 3 * hubs;
  double* acc = reinterpret_cast<double*>(hot + n_hot * PREP_STRIDE);
  int* grab_ctr = reinterpret_cast<int*>(acc + n_slots * 12);
  for (int i = threadIdx.x; i < n_slots * 12; i += E0C_BLOCK) acc[i] = 0;
  if (threadIdx.x == 0) *grab_ctr = 0;
  const double2* rec_img = reinterpret_cast<const double2*>(d.hot_rec);
  for (int i = threadIdx.x; i < n_hot * PREP_REC; i += E0C_BLOCK) {
    const int r = i / PREP_REC, j = i - r * PREP_REC;
    hot[r * PREP_STRIDE + j] = rec_img[(size_t)v.wg_cams[cam0 + r] * (HOT_REC_STRIDE / 2) + 6 + j];  // entries 12..23 of the image
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int t_begin = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t_end = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  auto grab = [&]() -> int {
    int n = 0;
    if (lane == 0) n = __hip_atomic_fetch_add(grab_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    n = __builtin_amdgcn_readfirstlane(n);
    const long long t = (long long)t_begin + n;
    return t < t_end ? (int)t : t_end;
  };
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)v.tile;
  auto tile_info = [&](int t, int& row0, int& k, int& nh, int& fl) {
    row0 = tiles[4 * t];
    k = tiles[4 * t + 1];
    nh = tiles[4 * t + 2];
    fl = tiles[4 * t + 3];
  };
  LplCursor pc;
  pc.t = grab();
  pc.pass = 0;
  pc.j = 0;
  pc.row0 = 0;
  pc.k = 1;
  int c_t = pc.t, c_row0 = 0, c_k = 0, c_nh = 0, c_fl = 0, nx_t = t_end;
  if (c_t < t_end) {
    tile_info(c_t, c_row0, c_k, c_nh, c_fl);
    pc.row0 = c_row0;
    pc.k = c_k;
    nx_t = grab();
  }
  auto issue = [&](LplRow& r) {
    if (pc.t < t_end) {
      const size_t i = ((size_t)pc.row0 + (pc.pass ? pc.k - 1 - pc.j : pc.j)) * WAVE + lane;
      r.uv = v.uv[i];
      r.cw = v.cw[i];
      if (ROBUST) r.w = v.w[i];
      if (++pc.j == pc.k) {
        pc.j = 0;
        if (++pc.pass == 2) {
          pc.pass = 0;
          pc.t = nx_t;
          if (pc.t < t_end) {
            int nh_, fl_;
            tile_info(pc.t, pc.row0, pc.k, nh_, fl_);
          }
        }
      }
    }
  };
  auto read_cam = [&](const double2* hp, Cam& P) {
    const double2 b0 = hp[0], b1 = hp[1], b2 = hp[2], b3 = hp[3], b4 = hp[4], b5 = hp[5];
    P.r0 = make_double4(b0.x, b0.y, b1.x, b1.y);
    P.r1 = make_double4(b2.x, b2.y, b3.x, b3.y);
    P.r2 = make_double4(b4.x, b4.y, b5.x, b5.y);
  };
  LplRow n1, n2, n3;
  n1.cw = n2.cw = n3.cw = -1;
  n1.w = n2.w = n3.w = 1.0;
  n1.uv = n2.uv = n3.uv = make_double2(0, 0);
  issue(n1);
  issue(n2);
  issue(n3);
  while (c_t < t_end) {
    const int lm = v.lm_of[(size_t)c_t * WAVE + lane];
    const int sg = v.seg[(size_t)c_t * WAVE + lane];
    const double4 X = v.lml[(size_t)c_t * WAVE + lane], s4 = v.lsc[(size_t)c_t * WAVE + lane];  // lane-ordered mirrors (V2)
    double hw[4], hbeta;
    house4(X, hw, hbeta);
    double red[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < c_k; ++j) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      Cam P;
      // LDS pointer or global pointer, never a select of the two (a generic pointer turns the reads into flat_loads)
      if (cur.cw >= 0) read_cam(hot + lpl_cw_slot(cur.cw) * PREP_STRIDE, P);
      else read_cam(rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2) + 6, P);
      const double sw = ROBUST ? sqrt(cur.w) : 1.0;
      const Hom h = hom_project(P, X, cur.uv.x, cur.uv.y);
      double jl4[8], jl3[6];
      hom_jl4(P, h, sw, s4, jl4);
      jl3_of_jl4(jl4, hw, hbeta, jl3);
      const double r0 = sw * h.r0, r1 = sw * h.r1;
      acc_h6(red, jl3);
#pragma unroll
      for (int m = 0; m < 3; ++m) red[6 + m] += jl3[m] * r0 + jl3[3 + m] * r1;
    }
    if (c_fl & 1) seg_reduce_steps<9>(red, lane, sg & 255, (sg >> 8) & 255, 4);
    double w3[3] = {0, 0, 0};
    if (lm >= 0) {
      double Hi[9];
      hinv_damped(red, d.lambda_lm, Hi);
      w3[0] = Hi[0] * red[6] + Hi[1] * red[7] + Hi[2] * red[8];
      w3[1] = Hi[3] * red[6] + Hi[4] * red[7] + Hi[5] * red[8];
      w3[2] = Hi[6] * red[6] + Hi[7] * red[7] + Hi[8] * red[8];
      // per-term record of this lane (e0_lpl_h); Hll^-1 and the row-major record of the other kernels once per landmark
      double* r2 = v.lmrec + ((size_t)c_t * LPL_REC_H) * WAVE + lane;
      const double rv[LPL_REC_H] = {X.x, X.y, X.z, X.w, s4.x, s4.y, s4.z, s4.w, Hi[0], Hi[1], Hi[2], Hi[4], Hi[5], Hi[8]};
#pragma unroll
      for (int m = 0; m < LPL_REC_H; ++m) r2[m * WAVE] = rv[m];
      if (lane == (sg & 255) && !d.prep_lpl_only) {
#pragma unroll
        for (int m = 0; m < 9; ++m) d.hll_inv[9 * (size_t)lm + m] = Hi[m];
        double4* rec = reinterpret_cast<double4*>(d.lmrec) + 4 * (size_t)lm;
        rec[0] = X;
        rec[1] = s4;
        rec[2] = make_double4(Hi[0], Hi[1], Hi[2], Hi[4]);
        rec[3] = make_double4(Hi[5], Hi[8], 0, 0);
      }
    }
    for (int jj = 0; jj < c_k; ++jj) {
      const int j = c_k - 1 - jj;
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      Cam P;
      // LDS pointer or global pointer, never a select of the two (a generic pointer turns the reads into flat_loads)
      if (cur.cw >= 0) read_cam(hot + lpl_cw_slot(cur.cw) * PREP_STRIDE, P);
      else read_cam(rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2) + 6, P);
      const double sw = ROBUST ? sqrt(cur.w) : 1.0;
      const Hom h = hom_project(P, X, cur.uv.x, cur.uv.y);
      double jl4[8], jl3[6];
      hom_jl4(P, h, sw, s4, jl4);
      jl3_of_jl4(jl4, hw, hbeta, jl3);
      const double e0 = sw * h.r0 - (jl3[0] * w3[0] + jl3[1] * w3[1] + jl3[2] * w3[2]);
      const double e1 = sw * h.r1 - (jl3[3] * w3[0] + jl3[4] * w3[1] + jl3[5] * w3[2]);
      const double4 q = hom_q(h, sw, e0, e1);
      if (cur.cw >= 0) {
        double* a = acc + lpl_acc_slot(cur.cw, hubs);
        const double val[12] = {X.x * q.x, X.y * q.x, X.z * q.x, X.w * q.x, X.x * q.y, X.y * q.y,
                                X.z * q.y, X.w * q.y, X.x * q.z, X.y * q.z, X.z * q.z, X.w * q.z};
#pragma unroll
        for (int m = 0; m < 12; ++m)
          __hip_atomic_fetch_add(a + m * n_slots, val[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        d.q4c[d.q_rows ? lpl_cold_q(c_fl, c_nh, j, lane) : v.cpos[((size_t)c_row0 + j) * WAVE + lane]] = q;
      }
    }
    c_t = nx_t;
    if (c_t < t_end) {
      tile_info(c_t, c_row0, c_k, c_nh, c_fl);
      nx_t = grab();
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_hot * 6; i += E0C_BLOCK) {
    const int r = i / 6, m = 2 * (i % 6);
    const double* a0 = acc + m * n_slots;
    const double* a1 = a0 + n_slots;
    double2 s;
    if (r < hubs) {
      s.x = (a0[4 * r] + a0[4 * r + 1]) + (a0[4 * r + 2] + a0[4 * r + 3]);
      s.y = (a1[4 * r] + a1[4 * r + 1]) + (a1[4 * r + 2] + a1[4 * r + 3]);
    } else {
      s.x = a0[r + 3 * hubs];
      s.y = a1[r + 3 * hubs];
    }
    reinterpret_cast<double2*>(hot_out + (size_t)v.wg_slot_rec[cam0 + r] * 12)[i % 6] = s;
  }
}

// b11_c = N_c^T y12_c for the per-camera sums of prepare_lpl_h (cam_cold_sum has applied sigma); y12 is scratch
// and left zeroed, as the dense-y term loop expects it
POVAR_KERNEL __launch_bounds__(256) void cam_nt_project(Dp d, double* y12, double* out11, const double* ncw) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.n_cams) return;
  double y[12], o[11];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    y[j] = y12[12 * (size_t)c + j];
    y12[12 * (size_t)c + j] = 0;
  }
  nt_apply(ncw + 13 * (size_t)c, ncw[13 * (size_t)c + 12], y, o);
#pragma unroll
  for (int j = 0; j < 11; ++j) out11[11 * (size_t)c + j] = o[j];
}



// K2' / K3' + K5' on the lane-per-landmark layout (the step-2 twins of lpl_pass, povar_kernels.hpp):
//   MODE 0  linearize_landmark_projective_space_homogeneous + scale_Jl_cols_homogeneous (landmark_block.hpp:180-225,
//           298-309) at (cams_lin4, lms_lin4): robust weight per observation (V2::w), the four Jl column scales per
//           landmark, finiteness flag; the per-slot arrays of the lane-per-observation kernels follow lazily.
//   MODE 1  compute_error_projective_space_homogeneous (helper.cpp:157-196) at (cams4, lms4): the six sums of OpErrorH
//           per workgroup into part[6 * blockIdx.x ..].
template <int MODE>
__global__ __launch_bounds__(E0C_BLOCK) void lpl_pass_h(Dp d, double* part) {
  extern __shared__ double2 hot[];  // [n_hot][PASS_REC] records (P_c row-major), then the tile counter
  __shared__ double sh[6 * (E0C_BLOCK / 64)];
  const V2& v = d.v2;
  const double4* cams = MODE == 0 ? d.cams_lin4 : d.cams4;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_hot = v.wg_cam_off[blockIdx.x + 1] - cam0;
  int* grab_ctr = reinterpret_cast<int*>(hot + n_hot * PASS_STRIDE);
  if (threadIdx.x == 0) *grab_ctr = 0;
  for (int i = threadIdx.x; i < n_hot * PASS_REC; i += E0C_BLOCK) {
    const int r = i / PASS_REC, j = i - r * PASS_REC;
    hot[r * PASS_STRIDE + j] = reinterpret_cast<const double2*>(cams + 3 * (size_t)d.hot_cams[v.wg_cams[cam0 + r]])[j];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int t_begin = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t_end = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  auto grab = [&]() -> int {
    int n = 0;
    if (lane == 0) n = __hip_atomic_fetch_add(grab_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    n = __builtin_amdgcn_readfirstlane(n);
    const long long t = (long long)t_begin + n;
    return t < t_end ? (int)t : t_end;
  };
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)v.tile;
  int c_t = grab(), q1 = c_t < t_end ? grab() : t_end, q2 = q1 < t_end ? grab() : t_end;
  int pc_t = c_t, pc_ahead = 0, pc_j = 0, pc_row0 = 0, pc_k = 1;
  if (pc_t < t_end) { pc_row0 = tiles[4 * pc_t]; pc_k = tiles[4 * pc_t + 1]; }
  auto issue = [&](LplRow& r) {
    if (pc_t < t_end) {
      const size_t i = ((size_t)pc_row0 + pc_j) * WAVE + lane;
      r.uv = v.uv[i];
      r.cw = v.cw[i];
      if (++pc_j == pc_k) {
        pc_j = 0;
        ++pc_ahead;
        pc_t = pc_ahead == 1 ? q1 : pc_ahead == 2 ? q2 : t_end;
        if (pc_t < t_end) { pc_row0 = tiles[4 * pc_t]; pc_k = tiles[4 * pc_t + 1]; }
      }
    }
  };
  LplRow n1, n2, n3;
  n1.cw = n2.cw = n3.cw = -1;
  n1.w = n2.w = n3.w = 1.0;
  n1.uv = n2.uv = n3.uv = make_double2(0, 0);
  issue(n1);
  issue(n2);
  issue(n3);
  double sc[6] = {0, 0, 0, 0, 0, 0};
  int bad = 0;
  while (c_t < t_end) {
    const int c_row0 = tiles[4 * c_t], c_k = tiles[4 * c_t + 1], c_fl = tiles[4 * c_t + 3];
    const double4 X = v.lmx[(size_t)c_t * WAVE + lane];
    if (MODE == 0) v.lml[(size_t)c_t * WAVE + lane] = X;  // the linearisation point, lane-ordered, is left behind
    double red[4] = {0, 0, 0, 0};
    for (int j = 0; j < c_k; ++j) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      // (a select of an LDS and a global pointer: six flat_loads.  Measured against ds_read / global_load in two branches
      // -- profiles/r02_ablations.txt item 16 --: the branches join with a wait on both counters, which drains the row
      // prefetch every step: 50 instead of 44 us here; the two-pass kernels with their longer steps gain from the split)
      const double2* hp = cur.cw >= 0 ? hot + lpl_cw_slot(cur.cw) * PASS_STRIDE
                                      : reinterpret_cast<const double2*>(cams + 3 * (size_t)d.hot_cams[-2 - cur.cw]);
      const double2 b0 = hp[0], b1 = hp[1], b2 = hp[2], b3 = hp[3], b4 = hp[4], b5 = hp[5];
      const Cam P = {make_double4(b0.x, b0.y, b1.x, b1.y), make_double4(b2.x, b2.y, b3.x, b3.y),
                     make_double4(b4.x, b4.y, b5.x, b5.y)};
      const Hom h = hom_project(P, X, cur.uv.x, cur.uv.y);
      const double r2 = h.r0 * h.r0 + h.r1 * h.r1;
      double e, w;
      error_weight(d, r2, e, w);
      if (MODE == 0) {
        const double sw = sqrt(w);
        bad |= !isfinite(r2) || !isfinite(sw) || !isfinite(h.D02) || !isfinite(h.D12);
        if (d.robust && v.w) v.w[((size_t)c_row0 + j) * WAVE + lane] = w;
        double jl[8];
        hom_jl4(P, h, sw, make_double4(1, 1, 1, 1), jl);
#pragma unroll
        for (int m = 0; m < 4; ++m) red[m] += jl[m] * jl[m] + jl[4 + m] * jl[4 + m];
      } else {
        bad |= !isfinite(r2);
        sc[0] += e; sc[1] += sqrt(r2); sc[2] += 1.0;
        if (h.valid) { sc[3] += e; sc[4] += sqrt(r2); sc[5] += 1.0; }
      }
    }
    if (MODE == 0) {
      const int sg = v.seg[(size_t)c_t * WAVE + lane];
      if (c_fl & 1) seg_reduce_steps<4>(red, lane, sg & 255, (sg >> 8) & 255, 4);
      // lane-ordered; the landmark-order copy (Dp::jl_scale4) is filled on demand (povar_lm.hip: ensure_jl_scale4)
      v.lsc[(size_t)c_t * WAVE + lane] = make_double4(1.0 / (d.eps + sqrt(red[0])), 1.0 / (d.eps + sqrt(red[1])),
                                                      1.0 / (d.eps + sqrt(red[2])), 1.0 / (d.eps + sqrt(red[3])));
    }
    c_t = q1;
    q1 = q2;
    q2 = q1 < t_end ? grab() : t_end;
    --pc_ahead;
  }
  if (bad) atomicOr(&d.flags[0], 1);
  if (MODE == 1) {
    block_sum<6, E0C_BLOCK>(sc, sh);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) part[6 * (size_t)blockIdx.x + k] = sc[k];
    }
  }
}

__host__ __device__ inline size_t back_lds_bytes_h(int n_hot) { return (size_t)n_hot * LPL_CAMREC_H * sizeof(double2) + 16; }

// K12' on the lane-per-landmark layout: back_substitute_joint (landmark_block.hpp:574-623) -- OpBackJoint's arithmetic
// on e0_lpl_h's records (z = sigma * N_c inc, left in the record image by cam_apply_inc_h, and P of the linearisation
// point).  First pass: H = Jl3^T Jl3 and Jl3^T (r + Jp inc); the lane solves for the damped increment and lifts it with
// the landmark's tangent basis; second pass: the model cost change, summed per workgroup into part[blockIdx.x].
template <bool ROBUST>
__global__ __launch_bounds__(E0C_BLOCK) void backsub_lpl_h(Dp d, double* part) {
  extern __shared__ double2 hot[];  // [n_hot][HOT_REC_H] records, then the tile counter
  __shared__ double sh[E0C_BLOCK / 64];
  const V2& v = d.v2;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_hot = v.wg_cam_off[blockIdx.x + 1] - cam0;
  int* grab_ctr = reinterpret_cast<int*>(hot + n_hot * LPL_CAMREC_H);
  if (threadIdx.x == 0) *grab_ctr = 0;
  const double2* rec_img = reinterpret_cast<const double2*>(d.hot_rec);
  for (int i = threadIdx.x; i < n_hot * HOT_REC_H; i += E0C_BLOCK)
    hot[(i / HOT_REC_H) * LPL_CAMREC_H + i % HOT_REC_H] =
        rec_img[(size_t)v.wg_cams[cam0 + i / HOT_REC_H] * (HOT_REC_STRIDE / 2) + i % HOT_REC_H];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int t_begin = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t_end = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  auto grab = [&]() -> int {
    int n = 0;
    if (lane == 0) n = __hip_atomic_fetch_add(grab_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    n = __builtin_amdgcn_readfirstlane(n);
    const long long t = (long long)t_begin + n;
    return t < t_end ? (int)t : t_end;
  };
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)v.tile;
  auto tile_info = [&](int t, int& row0, int& k, int& nh, int& fl) {
    row0 = tiles[4 * t];
    k = tiles[4 * t + 1];
    nh = tiles[4 * t + 2];
    fl = tiles[4 * t + 3];
  };
  LplCursor pc;
  pc.t = grab();
  pc.pass = 0;
  pc.j = 0;
  pc.row0 = 0;
  pc.k = 1;
  int c_t = pc.t, c_row0 = 0, c_k = 0, c_nh = 0, c_fl = 0, nx_t = t_end;
  if (c_t < t_end) {
    tile_info(c_t, c_row0, c_k, c_nh, c_fl);
    pc.row0 = c_row0;
    pc.k = c_k;
    nx_t = grab();
  }
  auto issue = [&](LplRow& r) {
    if (pc.t < t_end) {
      const size_t i = ((size_t)pc.row0 + (pc.pass ? pc.k - 1 - pc.j : pc.j)) * WAVE + lane;
      r.uv = v.uv[i];
      r.cw = v.cw[i];
      if (ROBUST) r.w = v.w[i];
      if (++pc.j == pc.k) {
        pc.j = 0;
        if (++pc.pass == 2) {
          pc.pass = 0;
          pc.t = nx_t;
          if (pc.t < t_end) {
            int nh_, fl_;
            tile_info(pc.t, pc.row0, pc.k, nh_, fl_);
          }
        }
      }
    }
  };
  // an observation's tile at the linearisation point and Jp * inc (z part of the record)
  auto obs = [&](const LplRow& cur, const double4& X, const double4& s4, const double (&hw)[4], double hbeta, Hom& h,
                 double (&jl4)[8], double (&jl3)[6], double& sw, double (&jpi)[2]) {
    // (select of an LDS and a global pointer: flat_loads; the split form spills here, see lpl_pass)
    const double2* hp = cur.cw >= 0 ? hot + lpl_cw_slot(cur.cw) * LPL_CAMREC_H
                                    : rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2);
    const double2 a0 = hp[0], a1 = hp[1], a2 = hp[2], a3 = hp[3], a4 = hp[4], a5 = hp[5];
    const double2 b0 = hp[6], b1 = hp[7], b2 = hp[8], b3 = hp[9], b4 = hp[10], b5 = hp[11];
    const double4 zz[3] = {make_double4(a0.x, a0.y, a1.x, a1.y), make_double4(a2.x, a2.y, a3.x, a3.y),
                           make_double4(a4.x, a4.y, a5.x, a5.y)};
    const Cam P = {make_double4(b0.x, b0.y, b1.x, b1.y), make_double4(b2.x, b2.y, b3.x, b3.y),
                   make_double4(b4.x, b4.y, b5.x, b5.y)};
    sw = ROBUST ? sqrt(cur.w) : 1.0;
    h = hom_project(P, X, cur.uv.x, cur.uv.y);
    hom_jl4(P, h, sw, s4, jl4);
    jl3_of_jl4(jl4, hw, hbeta, jl3);
    hom_jp_x(h, X, sw, zz, jpi);
  };
  LplRow n1, n2, n3;
  n1.cw = n2.cw = n3.cw = -1;
  n1.w = n2.w = n3.w = 1.0;
  n1.uv = n2.uv = n3.uv = make_double2(0, 0);
  issue(n1);
  issue(n2);
  issue(n3);
  double sc = 0;
  while (c_t < t_end) {
    const int lm = v.lm_of[(size_t)c_t * WAVE + lane];
    const int sg = v.seg[(size_t)c_t * WAVE + lane];
    const double4 X = v.lml[(size_t)c_t * WAVE + lane], s4 = v.lsc[(size_t)c_t * WAVE + lane];  // lane-ordered mirrors (V2)
    double hw[4], hbeta;
    house4(X, hw, hbeta);
    double red[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < c_k; ++j) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      Hom h;
      double jl4[8], jl3[6], sw, jpi[2];
      obs(cur, X, s4, hw, hbeta, h, jl4, jl3, sw, jpi);
      acc_h6(red, jl3);
      const double a0 = sw * h.r0 + jpi[0], a1 = sw * h.r1 + jpi[1];
#pragma unroll
      for (int m = 0; m < 3; ++m) red[6 + m] += jl3[m] * a0 + jl3[3 + m] * a1;
    }
    if (c_fl & 1) seg_reduce_steps<9>(red, lane, sg & 255, (sg >> 8) & 255, 4);
    double dp[4] = {0, 0, 0, 0};
    if (lm >= 0) {
      OpBackJoint::delta4(d, X, red, dp);
      double4 Xc = v.lmx[(size_t)c_t * WAVE + lane];
      Xc.x += dp[0] * s4.x; Xc.y += dp[1] * s4.y; Xc.z += dp[2] * s4.z; Xc.w += dp[3] * s4.w;
      v.lmx[(size_t)c_t * WAVE + lane] = Xc;  // the mirror stays current
      if (lane == (sg & 255)) d.lms4[lm] = Xc;
    }
    for (int jj = 0; jj < c_k; ++jj) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      Hom h;
      double jl4[8], jl3[6], sw, jpi[2];
      obs(cur, X, s4, hw, hbeta, h, jl4, jl3, sw, jpi);
      const double rr[2] = {sw * h.r0, sw * h.r1};
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double ji = jpi[r] + (jl4[4 * r] * dp[0] + jl4[4 * r + 1] * dp[1] + jl4[4 * r + 2] * dp[2] + jl4[4 * r + 3] * dp[3]);
        sc -= ji * (0.5 * ji + rr[r]);
      }
    }
    c_t = nx_t;
    if (c_t < t_end) {
      tile_info(c_t, c_row0, c_k, c_nh, c_fl);
      nx_t = grab();
    }
  }
  double sv[1] = {sc};
  block_sum<1, E0C_BLOCK>(sv, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = sv[0];
}

// cam_cold_sum fused with cam_binv_axpy_h (mode 2) for the unsharded LDSACC term loop of step 2
// (the step-2 twin of cam_cold_sum_binv): per-camera sum of the E0 row, tangent projection, B^-1 (11x11),
// AXPY and z = sigma * (N_c tmp) in one kernel.
template <int NT>
__global__ __launch_bounds__(NT) void cam_cold_sum_binv_h(Dp d, int want_norms, const double* ncw) {
  const int done = d.flags[1];
  __shared__ double sh[4 * 12];
  const int c = blockIdx.x, t = threadIdx.x;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0;
  const int2 pr = d.cmv.cam_range[c];
  const int p0 = pr.x, p1 = pr.y;
  const int r = d.hot_part ? d.cam_hot[c] : 0;
  const size_t base = 12 * (size_t)c;
  double bi[11], sg[12], w[12], acc_old = 0;
  const double beta = ncw[13 * (size_t)c + 12];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    sg[j] = d.sigma[base + j];
    w[j] = ncw[13 * (size_t)c + j];
  }
  if (t < 11) {
    const double* Bi = d.binv + 144 * (size_t)c + 11 * t;
#pragma unroll
    for (int j = 0; j < 11; ++j) bi[j] = Bi[j];
    acc_old = d.accum[11 * (size_t)c + t];
  }
  if (done) return;
  constexpr int U = 4;
  for (int pb = p0 + t; pb < p1; pb += U * NT) {
    double hx[U], hy[U], hz[U], hw[U];
    double4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pb + u * NT;
      const bool in = p < p1;
      const int pc = in ? p : p0;
      hx[u] = d.cmv.h[pc];
      hy[u] = d.cmv.h[d.cmv.n + pc];
      hz[u] = d.cmv.h[2 * d.cmv.n + pc];
      hw[u] = d.cmv.h[3 * d.cmv.n + pc];
      q[u] = in ? d.q4c[d.cmv.src ? d.cmv.src[pc] : pc] : make_double4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] += hx[u] * q[u].x; acc[1] += hy[u] * q[u].x; acc[2] += hz[u] * q[u].x; acc[3] += hw[u] * q[u].x;
      acc[4] += hx[u] * q[u].y; acc[5] += hy[u] * q[u].y; acc[6] += hz[u] * q[u].y; acc[7] += hw[u] * q[u].y;
      acc[8] += hx[u] * q[u].z; acc[9] += hy[u] * q[u].z; acc[10] += hz[u] * q[u].z; acc[11] += hw[u] * q[u].z;
    }
  }
  if (d.part_range) {  // e0_lpl_h: the camera's partial records are one contiguous run
    const int2 rr = d.part_range[c];
    for (int wg = rr.x + t; wg < rr.y; wg += NT) {
      const double* ip = d.hot_part + (size_t)wg * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] += ip[k];
    }
  } else if (r > 0 && r <= d.n_hot_acc) {
    for (int wg = t; wg < d.n_hot_wg; wg += NT) {
      const double* ip = d.hot_part + ((size_t)(r - 1) * d.n_hot_wg + wg) * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] += ip[k];
    }
  }
  block_sum_dpp<12, NT>(acc, sh);  // every thread now holds the 12 ambient sums
  if (t >= 64) return;
  double y[12], y11[11];
#pragma unroll
  for (int j = 0; j < 12; ++j) y[j] = acc[j] * sg[j];
  nt_apply(w, beta, y, y11);
  double s = 0;
  if (t < 11) {
#pragma unroll
    for (int j = 0; j < 11; ++j) s += bi[j] * y11[j];
  }
  double nrm[2] = {0, 0};
  if (t < 11) {
    const size_t idx = 11 * (size_t)c + t;
    const double a = acc_old + s;
    d.tmp[idx] = s;
    d.accum[idx] = a;
    nrm[0] = s * s;
    nrm[1] = a * a;
  }
  // p12 = N_c tmp11: p_i = [0; tmp]_i - beta w_i (w[1:] . tmp)
  double wsel = 0, wsel1 = 0, sgt = 0;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    wsel = (t == j) ? w[j] : wsel;
    sgt = (t == j) ? sg[j] : sgt;
    if (j > 0) wsel1 = (t == j - 1) ? w[j] : wsel1;
  }
  double wt = t < 11 ? wsel1 * s : 0.0;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) wt += shfl_xor_d(wt, m);  // lanes 0..15 hold the 11 products
  const double prev = shfl_up_d(s, 1);
  if (t < 12) {
    const double pa = (t == 0 ? 0.0 : prev) - beta * wsel * wt;
    store_z(d, c, t, pa * sgt);
  }
  if (want_norms) {
    wave_sum<2>(nrm);
    if (t == 0) {
      d.norm_part[2 * (size_t)c] = nrm[0];
      d.norm_part[2 * (size_t)c + 1] = nrm[1];
    }
  }
}

// K13' (linearizor_power_varproj.cpp:283-305) and the z = sigma * (N_c inc_c) needed by K12'.
// mode 1: z only (before back substitution); mode 2: P_c += reshape((N_c inc_c) * sigma)
POVAR_KERNEL __launch_bounds__(256) void cam_apply_inc_h(Dp d, int mode, const double* ncw) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.n_cams) return;
  const double* w = ncw + 13 * (size_t)c;
  const double beta = w[12];
  const double* x = d.inc + 11 * (size_t)c;
  double wt = 0;
  for (int j = 0; j < 11; ++j) wt += w[j + 1] * x[j];
  double* cams = reinterpret_cast<double*>(d.cams4) + 12 * (size_t)c;
  for (int i = 0; i < 12; ++i) {
    const double p = (i == 0 ? 0.0 : x[i - 1]) - beta * w[i] * wt;
    const double v = p * d.sigma[12 * (size_t)c + i];
    if (mode == 1) store_z(d, c, i, v);
    else cams[i] += v;
  }
}

// K15: P_c /= |P_c|_F, X_l /= X_l[3]  (bal_bundle_adjustment.cpp:700-705)
POVAR_KERNEL __launch_bounds__(256) void normalize_joint(Dp d, int64_t n_lanes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_lanes && d.v2.lm_of[i] >= 0) {  // lane-ordered mirror (V2::lmx), kept current
    double4 X = d.v2.lmx[i];
    const double w = X.w;
    X.x /= w; X.y /= w; X.z /= w; X.w /= w;
    d.v2.lmx[i] = X;
  }
  if (i < d.n_cams) {
    double* P = reinterpret_cast<double*>(d.cams4) + 12 * (size_t)i;
    double s = 0;
    for (int k = 0; k < 12; ++k) s += P[k] * P[k];
    s = sqrt(s);
    for (int k = 0; k < 12; ++k) P[k] /= s;
  }
  if (i < d.n_lms) {
    double4 X = d.lms4[i];
    const double w = X.w;
    X.x /= w; X.y /= w; X.z /= w; X.w /= w;
    d.lms4[i] = X;
  }
}

}  // namespace povar
