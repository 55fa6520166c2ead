// povar_kernels_sc.hpp -- the explicit-Schur-complement solvers of the reference's LinearizorSC
// (solver/linearizor_sc.cpp): --solver-type-step-1 PCG | CHOLESKY and --solver-type-step-2 RIPCG.
//
// The reference materialises the reduced camera matrix S = (Hpp + lambda I) - E0 as a hash map of
// 12x12 (11x11) blocks (cg/block_sparse_matrix.hpp:66-69, filled by add_Hb_pOSE / add_Hb_joint,
// sc/landmark_block.hpp:360-472, under an n_cams^2 mutex array) and runs a Ceres-style
// preconditioned CG on it (cg/conjugate_gradient.hpp:112-470) with the Schur-Jacobi preconditioner
// = inverse of the diagonal blocks of S (cg/preconditioner.hpp:66-135).  Here the operator is
// applied matrix-free -- S p = B p - E0 p with the same per-term E0 kernels the power series uses
// -- and only the block diagonal of S is formed:
//     S_cc = sigma ( sum_obs  T_i (x) h_i h_i^T ) sigma + lambda I,
//     T_i  = w C_i - w^2 C_i (P3 s Hll^-1 s P3^T) C_i          (step 1; C, P3 as in E0Core)
//     T_i  = w Dm_i^T (I_2 - Jl3_i Hll^-1 Jl3_i^T) Dm_i        (step 2; Dm = [[D00,0,D02],[0,D00,D12]])
// i.e. sixty moments per camera (3x3 symmetric T times the 10 entries of h h^T) next to the forty
// Gram moments of Hpp.  CHOLESKY assembles the dense upper triangle of S with the same closed forms.
#pragma once
#include "../../include/povar_hip.h"
#include "povar_kernels_joint.hpp"

namespace povar {

// device pointers of the explicit-SC solvers (allocated on first use)
struct ScP {
  double* dm_part;   // [n_items][60] per-item moments of the E0 block diagonal
  double* dm;        // [n_cams][60]
  double* bmat;      // [n_cams][144] B_c = Hpp_c + lambda I, dim x dim row-major
  double* minv;      // [n_cams][144] Schur-Jacobi preconditioner S_cc^-1, dim x dim row-major
  double* x;         // PCG vectors, [dim * n_cams] each
  double* r;
  double* p;
  double* q;
  double* zv;
  double* part;      // [n_cam_blocks][4] per-workgroup partial dot products
  double* s;         // scalars, PS_*
  const double* ncw; // step 2: Householder vector (12) + beta of every camera's tangent basis
};
enum { PS_RHO = 0, PS_BETA, PS_PQ, PS_ALPHA, PS_Q0, PS_Q1, PS_NORM_R, PS_NORM_B, PS_ZETA, PS_COUNT = 16 };

__device__ inline int sym6(int a, int b) {  // index of (a,b) in the packed upper 3x3
  if (a > b) { const int t = a; a = b; b = t; }
  return a * 3 - (a * (a - 1)) / 2 + (b - a);
}

// ------------------------------------------------------------------------------------------
// block diagonal of E0: per-item moments  sum_obs K_i (x) h_i h_i^T,
// K_i = Jp-structure^T (Jl_i Hll^-1 Jl_i^T) Jp-structure (3x3 symmetric), camera-major, one
// wavefront per item, fixed order (the i == j terms of landmark_block.hpp:388-399 / 452-463)
// ------------------------------------------------------------------------------------------
template <bool HOM>
__global__ __launch_bounds__(256) void cm_gram_sc(Dp d, double* part) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (item >= d.n_items) return;
  const int b = d.item_off[item], e = d.item_off[item + 1];
  const Cam P = load_cam(d.cams_lin4, d.item_cam[item]);
  double acc[60];
#pragma unroll
  for (int k = 0; k < 60; ++k) acc[k] = 0;
  for (int p = b + lane; p < e; p += WAVE) {
    const double sw = d.robust ? d.sw[d.cm_slot[p]] : 1.0;
    const double w = sw * sw;
    const double2 uv = d.cm_uv[p];
    const int lm = d.cm_lm[p];
    double K[6], hv[4];
    if (HOM) {
      const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 4 * (size_t)lm;
      const double4 X = rec[0], s = rec[1], r2 = rec[2], r3 = rec[3];
      const double Hi[9] = {r2.x, r2.y, r2.z, r2.y, r2.w, r3.x, r2.z, r3.x, r3.y};
      const Hom h = hom_project(P, X, uv.x, uv.y);
      double jl4[8], jl3[6], hw[4], beta;
      hom_jl4(P, h, sw, s, jl4);
      house4(X, hw, beta);
      jl3_of_jl4(jl4, hw, beta, jl3);
      double V[6];  // Jl3 Hll^-1 (2x3)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[3 * r + j] = jl3[3 * r] * Hi[j] + jl3[3 * r + 1] * Hi[3 + j] + jl3[3 * r + 2] * Hi[6 + j];
      const double W00 = V[0] * jl3[0] + V[1] * jl3[1] + V[2] * jl3[2];
      const double W01 = V[0] * jl3[3] + V[1] * jl3[4] + V[2] * jl3[5];
      const double W11 = V[3] * jl3[3] + V[4] * jl3[4] + V[5] * jl3[5];
      const double a = h.D00, b2 = h.D02, c2 = h.D12;
      K[0] = w * a * a * W00;
      K[1] = w * a * a * W01;
      K[2] = w * a * (b2 * W00 + c2 * W01);
      K[3] = w * a * a * W11;
      K[4] = w * a * (b2 * W01 + c2 * W11);
      K[5] = w * (b2 * b2 * W00 + 2.0 * b2 * c2 * W01 + c2 * c2 * W11);
      hv[0] = X.x; hv[1] = X.y; hv[2] = X.z; hv[3] = X.w;
    } else {
      const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 3 * (size_t)lm;
      const double4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
      const double s[3] = {r0.w, r1.x, r1.y};
      const double Hi[9] = {r1.z, r1.w, r2.x, r1.w, r2.y, r2.z, r2.x, r2.z, r2.w};
      const double P3[9] = {P.r0.x, P.r0.y, P.r0.z, P.r1.x, P.r1.y, P.r1.z, P.r2.x, P.r2.y, P.r2.z};
      double M[9], MH[9], A[9];  // M = P3 diag(s), A = M Hll^-1 M^T
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[3 * i + j] = P3[3 * i + j] * s[j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) MH[3 * i + j] = M[3 * i] * Hi[j] + M[3 * i + 1] * Hi[3 + j] + M[3 * i + 2] * Hi[6 + j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[3 * i + j] = MH[3 * i] * M[3 * j] + MH[3 * i + 1] * M[3 * j + 1] + MH[3 * i + 2] * M[3 * j + 2];
      const double sb2 = d.sb * d.sb;
      const double cu = sb2 * uv.x, cv = sb2 * uv.y, cuv = sb2 * (uv.x * uv.x + uv.y * uv.y);
      const double C[9] = {1, 0, -cu, 0, 1, -cv, -cu, -cv, cuv};
      double CA[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) CA[3 * i + j] = C[3 * i] * A[j] + C[3 * i + 1] * A[3 + j] + C[3 * i + 2] * A[6 + j];
      const double w2 = w * w;
      int k = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) K[k++] = w2 * (CA[3 * i] * C[j] + CA[3 * i + 1] * C[3 + j] + CA[3 * i + 2] * C[6 + j]);
      hv[0] = r0.x; hv[1] = r0.y; hv[2] = r0.z; hv[3] = 1.0;
    }
    const double hh[10] = {hv[0] * hv[0], hv[0] * hv[1], hv[0] * hv[2], hv[0] * hv[3], hv[1] * hv[1],
                           hv[1] * hv[2], hv[1] * hv[3], hv[2] * hv[2], hv[2] * hv[3], hv[3] * hv[3]};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[10 * k + j] += K[k] * hh[j];
  }
  wave_sum<60>(acc);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 60; ++k) part[60 * (size_t)item + k] = acc[k];
  }
}

// dm[c] = sum of the camera's item moments, fixed order: 16 wavefronts stride over the items (a hub
// camera of venice-1778 has ~800 of them), then a fixed-order sum of the 16 partials
POVAR_KERNEL __launch_bounds__(1024) void cam_sum_parts60(Dp d, const double* part, double* dm) {
  __shared__ double sh[16][60];
  const int c = blockIdx.x, e = threadIdx.x & 63, q = threadIdx.x >> 6;
  if (e < 60) {
    double s = 0;
    for (int it = d.cam_item_off[c] + q; it < d.cam_item_off[c + 1]; it += 16) s += part[60 * (size_t)it + e];
    sh[q][e] = s;
  }
  __syncthreads();
  if (threadIdx.x < 60) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sh[k][threadIdx.x];
    dm[60 * (size_t)c + threadIdx.x] = s;
  }
}

// Per camera: A = sigma (Hpp-moments [- E0-diagonal moments]) sigma, projected on the tangent space
// in step 2 (N_c^T A N_c), + lambda I.  out_mat receives A (dim x dim row-major), out_inv its
// inverse by Cholesky of the upper triangle (preconditioner.hpp:104-107 / LPV:145-148).
// One thread per camera, matrices in LDS [element][thread] like cam_build_binv.
constexpr int K8_SC_THREADS = 32;  // one thread per camera (explicit-SC path: not tuned further)
template <bool HOM>
__global__ __launch_bounds__(K8_SC_THREADS) void cam_build_sc(Dp d, double lambda, const double* ncw, const double* dm,
                                                           double* out_inv, double* out_mat) {
  __shared__ double A[144 * K8_SC_THREADS];
  __shared__ double X[144 * K8_SC_THREADS];
  const int t = threadIdx.x;
  const int c = blockIdx.x * K8_SC_THREADS + t;
  if (c >= d.n_cams) return;
  constexpr int DIM = HOM ? 11 : 12;
#define A_(i, j) A[((i) * 12 + (j)) * K8_SC_THREADS + t]
#define X_(i, j) X[((i) * 12 + (j)) * K8_SC_THREADS + t]
  const double* g = d.G + 40 * (size_t)c;
  const double* sg = d.sigma + 12 * (size_t)c;
  const double sb2 = d.sb * d.sb;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const int ij = sym10(i, j);
          double v;
          if (HOM) {
            if (a == b) v = a < 2 ? g[ij] : g[30 + ij];
            else if (a + b == 1) v = 0;
            else v = g[10 * ((a == 2 ? b : a) + 1) + ij];
          } else {
            if (a == b) v = a < 2 ? g[ij] : sb2 * g[30 + ij];
            else if (a + b == 1) v = 0;
            else v = -sb2 * g[10 * ((a == 2 ? b : a) + 1) + ij];
          }
          if (dm) v -= dm[60 * (size_t)c + 10 * sym6(a, b) + ij];
          A_(4 * a + i, 4 * b + j) = v * sg[4 * a + i] * sg[4 * b + j];
        }
  if (HOM) {
    const double* w = ncw + 13 * (size_t)c;
    const double beta = w[12];
    // T = A N (12 x 11) into X, then N^T T (11 x 11) back into A (row stride kept at 12)
    for (int i = 0; i < 12; ++i) {
      double aw = 0;
      for (int k = 0; k < 12; ++k) aw += A_(i, k) * w[k];
      for (int j = 0; j < 11; ++j) X_(i, j) = A_(i, j + 1) - beta * aw * w[j + 1];
    }
    for (int j = 0; j < 11; ++j) {
      double wt = 0;
      for (int k = 0; k < 12; ++k) wt += w[k] * X_(k, j);
      for (int i = 0; i < 11; ++i) A_(i, j) = X_(i + 1, j) - beta * w[i + 1] * wt;
    }
  }
  for (int j = 0; j < DIM; ++j) A_(j, j) += lambda;
  if (out_mat) {
    double* o = out_mat + 144 * (size_t)c;
    for (int i = 0; i < DIM; ++i)
      for (int j = 0; j < DIM; ++j) o[DIM * i + j] = A_(i, j);
  }
  if (out_inv) {
    for (int j = 0; j < DIM; ++j) {
      double dd = A_(j, j);
      for (int k = 0; k < j; ++k) dd -= A_(j, k) * A_(j, k);
      dd = sqrt(dd);
      A_(j, j) = dd;
      for (int i = j + 1; i < DIM; ++i) {
        double s = A_(j, i);
        for (int k = 0; k < j; ++k) s -= A_(i, k) * A_(j, k);
        A_(i, j) = s / dd;
      }
    }
    for (int col = 0; col < DIM; ++col) {
      for (int i = 0; i < DIM; ++i) {
        double s = (i == col) ? 1.0 : 0.0;
        for (int k = 0; k < i; ++k) s -= A_(i, k) * X_(k, col);
        X_(i, col) = s / A_(i, i);
      }
      for (int i = DIM - 1; i >= 0; --i) {
        double s = X_(i, col);
        for (int k = i + 1; k < DIM; ++k) s -= A_(k, i) * X_(k, col);
        X_(i, col) = s / A_(i, i);
      }
    }
    double* o = out_inv + 144 * (size_t)c;
    for (int i = 0; i < DIM; ++i)
      for (int j = 0; j < DIM; ++j) o[DIM * i + j] = X_(i, j);
  }
#undef A_
#undef X_
}

// ------------------------------------------------------------------------------------------
// PCG (conjugate_gradient.hpp:112-290 / 292-470, driven as linearizor_base.cpp:104-150):
// vectors are per-camera blocks of DIM doubles; one wavefront per camera, lane j < DIM owns
// component j; dot products go through per-workgroup partials and a one-wavefront scalar kernel,
// so the host reads one flag word per iteration.
// ------------------------------------------------------------------------------------------

// lane j < DIM: sum_k M[c][j][k] * v_k with v_k held by lane k
template <int DIM>
__device__ inline double block_row_times(const double* M, int c, int lane, double v) {
  double s = 0;
  const double* row = M + 144 * (size_t)c + DIM * (lane < DIM ? lane : 0);
#pragma unroll
  for (int k = 0; k < DIM; ++k) s += row[k] * shfl_d(v, k);
  return s;
}

// E0 input: z = sigma * v (step 1; plus v itself for the stored-tile forms) or sigma * (N_c v) (step 2)
// into the dense z + hot records
template <bool HOM>
__device__ inline void emit_z(const Dp& d, const double* ncw, int c, int lane, bool in, double v) {
  if (HOM) {
    const double* w = ncw + 13 * (size_t)(in ? c : 0);
    const double beta = w[12];
    double wt = (in && lane < 11) ? w[lane + 1] * v : 0.0;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) wt += shfl_xor_d(wt, m);  // lanes 0..15 hold the 11 products
    const double prev = shfl_up_d(v, 1);
    if (in && lane < 12) {
      const double pa = (lane == 0 ? 0.0 : prev) - beta * w[lane] * wt;
      store_z(d, c, lane, pa * d.sigma[12 * (size_t)c + lane]);
    }
  } else {
    if (in && lane < 12) {
      store_z(d, c, lane, v * d.sigma[12 * (size_t)c + lane]);
      d.tmp[12 * (size_t)c + lane] = v;  // the stored-tile E0 forms take the unscaled vector (sigma is in the tile)
    }
  }
}

template <int N>
__device__ inline void store_partials(const ScP& s, double (&v)[N], double* sh) {
  block_sum<N, K9_CAMS * 64>(v, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) s.part[4 * (size_t)blockIdx.x + k] = v[k];
  }
}

// x = 0, r = b, z = M^-1 r; partials (x.(b+r) = 0, r.r, r.z)     (CG:146-147, 163, 168-176)
template <int DIM>
__global__ __launch_bounds__(K9_CAMS * 64) void pcg_init(Dp d, ScP s) {
  __shared__ double sh[K9_CAMS * 3];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * K9_CAMS + (threadIdx.x >> 6);
  const bool in = c < d.n_cams, act = in && lane < DIM;
  const size_t idx = (size_t)DIM * (in ? c : 0) + (lane < DIM ? lane : 0);
  const double r = act ? d.b[idx] : 0.0;
  const double z = block_row_times<DIM>(s.minv, in ? c : 0, lane, r);
  double v[3] = {0, 0, 0};
  if (act) {
    s.x[idx] = 0;
    s.r[idx] = r;
    s.zv[idx] = z;
    v[1] = r * r;
    v[2] = r * z;
  }
  store_partials<3>(s, v, sh);
}

// end of iteration `it` (it == 0: after pcg_init): the termination tests of CG:243-301 and, when
// the loop goes on, rho / beta of the next iteration with their failure exits (CG:175-197)
POVAR_KERNEL __launch_bounds__(64) void pcg_check(Dp d, ScP s, int n_blocks, int it, int min_it, int max_it, double eta,
                                                double r_tol) {
  if (d.flags[1]) return;
  double v[3] = {0, 0, 0};
  for (int k = threadIdx.x; k < n_blocks; k += 64) {
    v[0] += s.part[4 * (size_t)k];
    v[1] += s.part[4 * (size_t)k + 1];
    v[2] += s.part[4 * (size_t)k + 2];
  }
  wave_sum<3>(v);
  if (threadIdx.x != 0) return;
  double* S = s.s;
  bool done = false;
  int status = POVAR_LINEAR_SOLVER_NO_CONVERGENCE, iters = it;
  const double norm_r = sqrt(v[1]);
  if (it == 0) {
    S[PS_NORM_B] = norm_r;
    S[PS_Q0] = 0.0;    // -x.(b + r) at x = 0
    S[PS_RHO] = 1.0;   // CG:160
    if (norm_r == 0.0) { done = true; status = POVAR_LINEAR_SOLVER_SUCCESS; }                        // CG:131-136
    else if (min_it == 0 && norm_r <= r_tol * norm_r) { done = true; status = POVAR_LINEAR_SOLVER_SUCCESS; }  // CG:149-158
  } else {
    const double q1 = -1.0 * v[0];
    const double zeta = it * (q1 - S[PS_Q0]) / q1;  // CG:268
    S[PS_Q1] = q1;
    S[PS_ZETA] = zeta;
    S[PS_NORM_R] = norm_r;
    if (zeta < eta && it >= min_it) { done = true; status = POVAR_LINEAR_SOLVER_SUCCESS; }            // CG:269-282
    else {
      S[PS_Q0] = q1;
      if (norm_r <= r_tol * S[PS_NORM_B] && it >= min_it) { done = true; status = POVAR_LINEAR_SOLVER_SUCCESS; }  // CG:288-297
      else if (it >= max_it) done = true;                                                            // CG:299-301
    }
  }
  if (!done) {
    const double last_rho = S[PS_RHO], rho = v[2];
    S[PS_RHO] = rho;
    if (rho == 0.0 || isinf(rho)) { done = true; status = POVAR_LINEAR_SOLVER_FAILURE; iters = it + 1; }  // CG:177-185
    else if (it >= 1) {
      const double beta = rho / last_rho;
      S[PS_BETA] = beta;
      if (beta == 0.0 || isinf(beta)) { done = true; status = POVAR_LINEAR_SOLVER_FAILURE; iters = it + 1; }  // CG:190-197
    }
  }
  if (done) {
    d.flags[1] = 1;
    d.flags[2] = iters;
    d.flags[3] = status;
  }
}

// p = z (first) or z + beta p; E0 input z-buffer = sigma * (N) p                       (CG:187-199)
template <int DIM, bool HOM>
__global__ __launch_bounds__(K9_CAMS * 64) void pcg_dir(Dp d, ScP s, int first) {
  if (d.flags[1]) return;
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * K9_CAMS + (threadIdx.x >> 6);
  const bool in = c < d.n_cams, act = in && lane < DIM;
  const size_t idx = (size_t)DIM * (in ? c : 0) + (lane < DIM ? lane : 0);
  double p = 0;
  if (act) {
    p = first ? s.zv[idx] : s.zv[idx] + s.s[PS_BETA] * s.p[idx];
    s.p[idx] = p;
  }
  emit_z<HOM>(d, s.ncw, c, lane, in, p);
}

// res = B v - E0 v with E0 v in the dense ambient y (sigma applied); lane j < DIM gets component j
template <int DIM, bool HOM>
__device__ inline double schur_times(const Dp& d, const ScP& s, int c, int lane, bool in, double v) {
  const double bv = block_row_times<DIM>(s.bmat, in ? c : 0, lane, v);
  double e0;
  if (HOM) {
    double y[12], y11[11];
#pragma unroll
    for (int j = 0; j < 12; ++j) y[j] = d.y[12 * (size_t)(in ? c : 0) + j];
    nt_apply(s.ncw + 13 * (size_t)(in ? c : 0), s.ncw[13 * (size_t)(in ? c : 0) + 12], y, y11);
    e0 = 0;
#pragma unroll
    for (int j = 0; j < 11; ++j) e0 = (lane == j) ? y11[j] : e0;
  } else {
    e0 = d.y[12 * (size_t)(in ? c : 0) + (lane < 12 ? lane : 0)];
  }
  return bv - e0;
}

// q = S p, partial p.q                                                                    (CG:201-202)
template <int DIM, bool HOM>
__global__ __launch_bounds__(K9_CAMS * 64) void pcg_apply(Dp d, ScP s) {
  if (d.flags[1]) return;
  __shared__ double sh[K9_CAMS];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * K9_CAMS + (threadIdx.x >> 6);
  const bool in = c < d.n_cams, act = in && lane < DIM;
  const size_t idx = (size_t)DIM * (in ? c : 0) + (lane < DIM ? lane : 0);
  const double p = act ? s.p[idx] : 0.0;
  const double q = schur_times<DIM, HOM>(d, s, c, lane, in, p);
  double v[1] = {0};
  if (act) {
    s.q[idx] = q;
    v[0] = p * q;
  }
  store_partials<1>(s, v, sh);
}

// alpha = rho / p.q with the exits of CG:204-223
POVAR_KERNEL __launch_bounds__(64) void pcg_alpha(Dp d, ScP s, int n_blocks, int it) {
  if (d.flags[1]) return;
  double v[1] = {0};
  for (int k = threadIdx.x; k < n_blocks; k += 64) v[0] += s.part[4 * (size_t)k];
  wave_sum<1>(v);
  if (threadIdx.x != 0) return;
  const double pq = v[0];
  s.s[PS_PQ] = pq;
  int status = -1;
  if (pq <= 0 || isinf(pq)) status = POVAR_LINEAR_SOLVER_NO_CONVERGENCE;  // "Matrix is indefinite" CG:204-215
  else {
    const double alpha = s.s[PS_RHO] / pq;
    s.s[PS_ALPHA] = alpha;
    if (isinf(alpha)) status = POVAR_LINEAR_SOLVER_FAILURE;  // CG:218-223
  }
  if (status >= 0) {
    d.flags[1] = 1;
    d.flags[2] = it;
    d.flags[3] = status;
  }
}

// x += alpha p.  reset == 0: r -= alpha q, z = M^-1 r and the partials of the iteration-end tests
// (CG:225, 237-239, 243, 286).  reset == 1 (every residual_reset_period-th iteration, CG:234-236):
// only x, and x goes to the E0 input so that pcg_residual can recompute r = b - S x.
template <int DIM, bool HOM>
__global__ __launch_bounds__(K9_CAMS * 64) void pcg_update(Dp d, ScP s, int reset) {
  if (d.flags[1]) return;
  __shared__ double sh[K9_CAMS * 3];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * K9_CAMS + (threadIdx.x >> 6);
  const bool in = c < d.n_cams, act = in && lane < DIM;
  const size_t idx = (size_t)DIM * (in ? c : 0) + (lane < DIM ? lane : 0);
  const double alpha = s.s[PS_ALPHA];
  double x = 0, r = 0;
  if (act) {
    x = s.x[idx] + alpha * s.p[idx];
    s.x[idx] = x;
  }
  if (reset) {
    emit_z<HOM>(d, s.ncw, c, lane, in, x);
    return;
  }
  if (act) {
    r = s.r[idx] - alpha * s.q[idx];
    s.r[idx] = r;
  }
  const double z = block_row_times<DIM>(s.minv, in ? c : 0, lane, r);
  double v[3] = {0, 0, 0};
  if (act) {
    s.zv[idx] = z;
    v[0] = x * (d.b[idx] + r);
    v[1] = r * r;
    v[2] = r * z;
  }
  store_partials<3>(s, v, sh);
}

// r = b - S x after pcg_update(reset = 1) and E0 x; z = M^-1 r; iteration-end partials   (CG:234-236)
template <int DIM, bool HOM>
__global__ __launch_bounds__(K9_CAMS * 64) void pcg_residual(Dp d, ScP s) {
  if (d.flags[1]) return;
  __shared__ double sh[K9_CAMS * 3];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * K9_CAMS + (threadIdx.x >> 6);
  const bool in = c < d.n_cams, act = in && lane < DIM;
  const size_t idx = (size_t)DIM * (in ? c : 0) + (lane < DIM ? lane : 0);
  const double x = act ? s.x[idx] : 0.0;
  const double sx = schur_times<DIM, HOM>(d, s, c, lane, in, x);
  const double bb = act ? d.b[idx] : 0.0;
  const double r = act ? bb - sx : 0.0;
  const double z = block_row_times<DIM>(s.minv, in ? c : 0, lane, r);
  double v[3] = {0, 0, 0};
  if (act) {
    s.r[idx] = r;
    s.zv[idx] = z;
    v[0] = x * (bb + r);
    v[1] = r * r;
    v[2] = r * z;
  }
  store_partials<3>(s, v, sh);
}

// inc = -x ("we solve H(-x) = b", linearizor_base.cpp:121-122) into the increment buffer
POVAR_KERNEL __launch_bounds__(256) void pcg_finish(const double* x, double* accum, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) accum[i] = -x[i];
}

// ------------------------------------------------------------------------------------------
// CHOLESKY (solve_direct_pOSE, sc/linearization_sc.hpp:236-245): dense S, row-major n x n with
// n = 12 n_cams; only the upper block triangle (camera_i <= camera_j) is formed, which is what the
// factorisation reads.  Off-diagonal part of add_Hb_pOSE (landmark_block.hpp:388-399) in closed form:
//     block(ci, cj) -= sigma_ci [ (F_i Hll^-1 F_j^T) (x) h h^T ] sigma_cj,   F_i = w_i C_i P3_i diag(s)
// One workgroup per landmark, observations staged 64 at a time in LDS, 16-lane groups take one
// camera pair each and add its 144 entries with fp64 atomics (hub cameras' blocks are shared by many
// landmarks).  S is the augmented row-major matrix of povar_kernels_chol.hpp (row stride ld).
// ------------------------------------------------------------------------------------------
constexpr int SCD_CHUNK = 64;
struct ScdObs {
  double F[9], G[9];  // F_i and F_i Hll^-1
  int cam;
};
__device__ inline void scd_stage(const Dp& d, int slot, const double* s3, const double* Hi, ScdObs& o) {
  const int cam = d.cam[slot];
  const Cam P = load_cam(d.cams_lin4, cam);
  const double2 uv = d.uv[slot];
  const double sw = d.robust ? d.sw[slot] : 1.0;
  const double w = sw * sw;
  const double sb2 = d.sb * d.sb;
  const double cu = sb2 * uv.x, cv = sb2 * uv.y, cuv = sb2 * (uv.x * uv.x + uv.y * uv.y);
  const double C[9] = {1, 0, -cu, 0, 1, -cv, -cu, -cv, cuv};
  const double P3[9] = {P.r0.x, P.r0.y, P.r0.z, P.r1.x, P.r1.y, P.r1.z, P.r2.x, P.r2.y, P.r2.z};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o.F[3 * i + j] = w * (C[3 * i] * P3[j] + C[3 * i + 1] * P3[3 + j] + C[3 * i + 2] * P3[6 + j]) * s3[j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o.G[3 * i + j] = o.F[3 * i] * Hi[j] + o.F[3 * i + 1] * Hi[3 + j] + o.F[3 * i + 2] * Hi[6 + j];
  o.cam = cam;
}

POVAR_KERNEL __launch_bounds__(256) void sc_dense_offdiag(Dp d, const int* lm_slot0, const int* lm_cnt, double* S, int64_t ld) {
  __shared__ ScdObs oi[SCD_CHUNK], oj[SCD_CHUNK];
  const int lm = blockIdx.x;
  const int s0 = lm_slot0[lm], k = lm_cnt[lm];
  const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 3 * (size_t)lm;
  const double4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
  const double h[4] = {r0.x, r0.y, r0.z, 1.0};
  const double s3[3] = {r0.w, r1.x, r1.y};
  const double Hi[9] = {r1.z, r1.w, r2.x, r1.w, r2.y, r2.z, r2.x, r2.z, r2.w};
  const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
  for (int ic = 0; ic < k; ic += SCD_CHUNK) {
    const int ni = min(SCD_CHUNK, k - ic);
    __syncthreads();
    if ((int)threadIdx.x < ni) scd_stage(d, s0 + ic + threadIdx.x, s3, Hi, oi[threadIdx.x]);
    for (int jc = ic; jc < k; jc += SCD_CHUNK) {
      const int nj = min(SCD_CHUNK, k - jc);
      __syncthreads();
      if ((int)threadIdx.x < nj) scd_stage(d, s0 + jc + threadIdx.x, s3, Hi, oj[threadIdx.x]);
      __syncthreads();
      for (int pp = grp; pp < ni * nj; pp += 16) {
        const int i = pp / nj, j = pp - i * nj;
        if (ic == jc && i > j) continue;  // cameras ascend inside a landmark: keep camera_i <= camera_j
        const int ci = oi[i].cam, cj = oj[j].cam;
        const double* sgi = d.sigma + 12 * (size_t)ci;
        const double* sgj = d.sigma + 12 * (size_t)cj;
        for (int e = gl; e < 144; e += 16) {
          const int r = e / 12, c = e - 12 * r;
          const int a = r >> 2, ii = r & 3, b = c >> 2, jj = c & 3;
          const double t = oi[i].G[3 * a] * oj[j].F[3 * b] + oi[i].G[3 * a + 1] * oj[j].F[3 * b + 1] +
                           oi[i].G[3 * a + 2] * oj[j].F[3 * b + 2];
          const double v = -t * h[ii] * h[jj] * sgi[r] * sgj[c];
          atomicAdd(S + (12 * (int64_t)ci + r) * ld + 12 * (int64_t)cj + c, v);
        }
      }
    }
  }
}

// S_cc += B_c = Hpp_c + lambda I (landmark_block.hpp:381-384 + linearization_sc.hpp:477-481) and the
// right-hand side -b into column N of the augmented matrix (povar_kernels_chol.hpp)
POVAR_KERNEL __launch_bounds__(256) void sc_dense_diag(Dp d, const double* bmat, double* S, int64_t ld, int N) {
  const int c = blockIdx.x, e = threadIdx.x;
  if (e < 144) {
    const int r = e / 12, cc = e - 12 * r;
    atomicAdd(S + (12 * (int64_t)c + r) * ld + 12 * (int64_t)c + cc, bmat[144 * (size_t)c + e]);
  }
  if (e < 12) S[(12 * (int64_t)c + e) * ld + N] = -d.b[12 * (size_t)c + e];
}

}  // namespace povar
