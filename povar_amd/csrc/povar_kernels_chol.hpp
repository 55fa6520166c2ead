// povar_kernels_chol.hpp -- dense fp64 Cholesky solve of the reduced camera system for
// --solver-type-step-1 CHOLESKY (solve_direct_pOSE, sc/linearization_sc.hpp:236-245, where the
// reference calls Eigen::SimplicialLLT on the assembled sparse S).
//
// Layout: M is row-major (N + 64) x LD, N = 12 n_cams rounded up to 64, LD = N + 128.  Columns [0, N)
// of rows [0, N) hold the upper triangle of S (identity on the padding rows), column N holds the
// right-hand side -b; the remaining rows / columns are slack that the 128 x 128 update tiles may touch.
// Factorisation S = R^T R (R upper, row-major), right-looking, two levels of blocking: panels of 64
// rows inside blocks of 256 rows,
//   chol_diag        R11 = chol(S11)                            one workgroup, block in LDS
//   chol_trsm        R12 = R11^-T S12 (and y_k = R11^-T rhs_k)    one thread per column
//   chol_syrk        rows of the same 256-block -= R12^T R12     64 x 64 tiles, K = 64
//   chol_syrk_outer  rows below the block -= P^T P, P = the block's 256 finished rows
//                                                               128 x 128 tiles, K = 256
// Carrying the right-hand side as column N folds the forward substitution into the factorisation;
// chol_back_* then solve R x = y panel by panel from the bottom.
// The trailing update is the n^3/3 part (venice-1778: 3.2 TFLOP): the only GEMM-shaped work of
// the whole path, so it runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64); both operands of
// P^T P are rows of the same row panel, so the A and B fragments are plain coalesced row reads staged
// in LDS.  It is bound by memory traffic, not by the matrix cores: with K = 64 every panel re-reads and
// re-writes the whole trailing matrix (8 flop per byte of C: 400 GB on venice-1778); K = 256 and
// 128 x 128 tiles cut the C traffic 4x and the operand traffic 2x (profiles/r01_g_*).
#pragma once
#include "povar_kernels.hpp"

namespace povar {

constexpr int CH_NB = 64;    // panel height == inner tile edge
constexpr int CH_OB = 256;   // outer block height (K of the big trailing update)
constexpr int CH_T = 128;    // outer tile edge
constexpr int CH_KC = 32;    // K chunk of the outer update staged in LDS
constexpr int CH_TLDS = CH_T + 16;  // LDS row stride (doubles): consecutive rows 32 banks apart
constexpr int CH_LDS = 80;   // LDS row stride in doubles: rows 2 apart share banks, the 32-lane halves of a b64 read do not

typedef double ch_d4 __attribute__((ext_vector_type(4)));

// R11 = chol(S11): upper Cholesky of the 64 x 64 diagonal block at (k0, k0); info |= 1 when a pivot
// is not positive.  One wavefront, lane c keeps column c in registers; row j of R reaches the other
// lanes through constant-lane broadcasts, so the 64 dependent steps need no LDS and no barriers
// (the 256-thread LDS version took 74 us per block, profiles/r01_g_*).
POVAR_KERNEL __launch_bounds__(64) void chol_diag(double* M, int64_t ld, int k0, int* info) {
  const int c = threadIdx.x;
  double* blk = M + (int64_t)k0 * ld + k0;
  double a[CH_NB];
#pragma unroll
  for (int r = 0; r < CH_NB; ++r) a[r] = blk[(int64_t)r * ld + c];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < CH_NB; ++j) {
    const double piv = bcast_lane(a[j], j);
    bad |= !(piv > 0);
    const double d = sqrt(piv);
    const double rjc = c < j ? 0.0 : (c == j ? d : a[j] / d);
    a[j] = rjc;
#pragma unroll
    for (int r = j + 1; r < CH_NB; ++r) a[r] -= bcast_lane(rjc, r) * rjc;  // A[r][c] -= R[j][r] R[j][c]
  }
  if (c == 0 && bad) atomicOr(info, 1);
#pragma unroll
  for (int r = 0; r < CH_NB; ++r) blk[(int64_t)r * ld + c] = c >= r ? a[r] : 0.0;
}

// R12 = R11^-T S12 for the columns [k0 + 64, N]: forward substitution down each column, one thread
// per column, R11 broadcast from LDS (reading R11 with wave-uniform scalar loads instead was slower:
// 58 vs 46 us per panel on venice-1778)
POVAR_KERNEL __launch_bounds__(128) void chol_trsm(double* M, int64_t ld, int k0) {
  __shared__ double R[CH_NB][CH_NB];
  const double* blk = M + (int64_t)k0 * ld + k0;
  for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 128) R[e >> 6][e & 63] = blk[(int64_t)(e >> 6) * ld + (e & 63)];
  __syncthreads();
  const int64_t c = (int64_t)k0 + CH_NB + (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (c >= ld) return;  // (the launch covers the columns through the rhs column N)
  double* col = M + (int64_t)k0 * ld + c;
  double s[CH_NB];
#pragma unroll
  for (int i = 0; i < CH_NB; ++i) s[i] = col[(int64_t)i * ld];
#pragma unroll
  for (int p = 0; p < CH_NB; ++p) {
    const double y = s[p] / R[p][p];
    s[p] = y;
#pragma unroll
    for (int i = p + 1; i < CH_NB; ++i) s[i] -= R[p][i] * y;
  }
#pragma unroll
  for (int i = 0; i < CH_NB; ++i) col[(int64_t)i * ld] = s[i];
}

// Rows of the current 256-block below the panel at k0: C -= R12^T R12 on the upper block triangle
// (including the rhs tile).  grid = (n_jt, n_it) with n_it limited to the block's remaining panels;
// tile (it, jt) with jt >= it covers rows k1 + 64 it, columns k1 + 64 jt.
// 4 wavefronts per workgroup, each a 32 x 32 quadrant = 2 x 2 MFMA tiles; fragments:
// A (16x4): lane l holds panel[kk + (l >> 4)][i + (l & 15)], B likewise with j, D: col = l & 15,
// row = (l >> 4) + 4 reg.
POVAR_KERNEL __launch_bounds__(256) void chol_syrk(double* M, int64_t ld, int k0) {
  const int it = blockIdx.y, jt = blockIdx.x;
  if (jt < it) return;
  __shared__ double As[CH_NB][CH_LDS];
  __shared__ double Bs[CH_NB][CH_LDS];
  const int k1 = k0 + CH_NB;
  const int64_t i0 = (int64_t)k1 + (int64_t)it * CH_NB, j0 = (int64_t)k1 + (int64_t)jt * CH_NB;
  const double* pan = M + (int64_t)k0 * ld;
  for (int e = threadIdx.x; e < CH_NB * CH_NB / 2; e += 256) {
    const int p = e >> 5, q = (e & 31) * 2;
    const double2 a = *reinterpret_cast<const double2*>(pan + (int64_t)p * ld + i0 + q);
    const double2 b = *reinterpret_cast<const double2*>(pan + (int64_t)p * ld + j0 + q);
    As[p][q] = a.x; As[p][q + 1] = a.y;
    Bs[p][q] = b.x; Bs[p][q + 1] = b.y;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wi = (w >> 1) * 32, wj = (w & 1) * 32;
  const int lr = lane >> 4, lc = lane & 15;
  ch_d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = ch_d4{0, 0, 0, 0};
#pragma unroll 4
  for (int kk = 0; kk < CH_NB; kk += 4) {
    const double a0 = As[kk + lr][wi + lc], a1 = As[kk + lr][wi + 16 + lc];
    const double b0 = Bs[kk + lr][wj + lc], b1 = Bs[kk + lr][wj + 16 + lc];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* cp = M + (i0 + wi + 16 * x + lr + 4 * r) * ld + j0 + wj + 16 * y + lc;
        *cp -= acc[x][y][r];
      }
}

// Rows below a finished block of `kdepth` rows (P = M[K0 : K0 + kdepth, :]):  C -= P^T P on the upper
// block triangle starting at row/column R0 = K0 + kdepth, including the rhs column.  grid = (n_jt, n_it),
// 128 x 128 tiles, 4 wavefronts each a 64 x 64 quadrant = 4 x 4 MFMA tiles; K walked in chunks of 32
// (2 x 36 KiB of LDS: two workgroups per CU overlap each other's loads and MFMAs).
POVAR_KERNEL __launch_bounds__(256, 2) void chol_syrk_outer(double* M, int64_t ld, int K0, int kdepth) {
  const int it = blockIdx.y, jt = blockIdx.x;
  if (jt < it) return;
  __shared__ double As[CH_KC][CH_TLDS];
  __shared__ double Bs[CH_KC][CH_TLDS];
  const int64_t R0 = (int64_t)K0 + kdepth;
  const int64_t i0 = R0 + (int64_t)it * CH_T, j0 = R0 + (int64_t)jt * CH_T;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wi = (w >> 1) * 64, wj = (w & 1) * 64;
  const int lr = lane >> 4, lc = lane & 15;
  ch_d4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = ch_d4{0, 0, 0, 0};
  // software pipeline: the global loads of chunk kc + 1 are in flight while the MFMAs of chunk kc run
  constexpr int NLD = CH_KC * CH_T / 2 / 256;  // double2 per thread and operand
  double2 ra[NLD], rb[NLD];
  auto fetch = [&](int kc) {
    const double* pan = M + ((int64_t)K0 + kc) * ld;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int e = threadIdx.x + 256 * u;
      const int p = e >> 6, q = (e & 63) * 2;
      ra[u] = *reinterpret_cast<const double2*>(pan + (int64_t)p * ld + i0 + q);
      rb[u] = *reinterpret_cast<const double2*>(pan + (int64_t)p * ld + j0 + q);
    }
  };
  fetch(0);
  for (int kc = 0; kc < kdepth; kc += CH_KC) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int e = threadIdx.x + 256 * u;
      const int p = e >> 6, q = (e & 63) * 2;
      As[p][q] = ra[u].x; As[p][q + 1] = ra[u].y;
      Bs[p][q] = rb[u].x; Bs[p][q + 1] = rb[u].y;
    }
    __syncthreads();
    if (kc + CH_KC < kdepth) fetch(kc + CH_KC);
#pragma unroll 2
    for (int kk = 0; kk < CH_KC; kk += 4) {
      double a[4], b[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        a[x] = As[kk + lr][wi + 16 * x + lc];
        b[x] = Bs[kk + lr][wj + 16 * x + lc];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
    }
  }
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* cp = M + (i0 + wi + 16 * x + lr + 4 * r) * ld + j0 + wj + 16 * y + lc;
        *cp -= acc[x][y][r];
      }
}

// back substitution R x = y, panel by panel from the bottom, in place on x (initialised with y =
// column N of M by chol_copy_rhs):
//   chol_back_solve   x_k = R11^-1 x_k                      one wavefront, 64 x 64 triangle
//   chol_back_update  x[0, k0) -= R[0:k0, k0:k0+64] x_k     one wavefront per row, 512-byte row reads
// so the upper triangle is streamed once over the whole chip (a single workgroup walking the
// row panels took 340 us per panel on venice-1778, profiles/r01_g_*).
POVAR_KERNEL __launch_bounds__(256) void chol_copy_rhs(const double* M, int64_t ld, int N, double* x) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N) x[i] = M[(int64_t)i * ld + N];
}

POVAR_KERNEL __launch_bounds__(64) void chol_back_solve(const double* M, int64_t ld, int k0, double* x) {
  const int lane = threadIdx.x;
  const double* blk = M + (int64_t)k0 * ld + k0;
  double u[CH_NB];  // column `lane` of R11
#pragma unroll
  for (int r = 0; r < CH_NB; ++r) u[r] = blk[(int64_t)r * ld + lane];
  const double t = x[k0 + lane];
  double xi = 0;  // lane p ends up holding x[k0 + p]
#pragma unroll
  for (int i = CH_NB - 1; i >= 0; --i) {
    double s[1] = {lane > i ? u[i] * xi : 0.0};  // sum_{p > i} R[i][p] x_p
    wave_sum<1>(s);
    const double v = (bcast_lane(t, i) - s[0]) / bcast_lane(u[i], i);
    if (lane == i) xi = v;
  }
  x[k0 + lane] = xi;
}

POVAR_KERNEL __launch_bounds__(256) void chol_back_update(const double* M, int64_t ld, int k0, double* x) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= k0) return;
  double s[1] = {M[(int64_t)row * ld + k0 + lane] * x[k0 + lane]};
  wave_sum<1>(s);
  if (lane == 0) x[row] -= s[0];
}

// identity on the padding rows [n, N) of the augmented matrix (after the memset)
POVAR_KERNEL __launch_bounds__(64) void chol_pad(double* M, int64_t ld, int n, int N) {
  const int r = n + threadIdx.x;
  if (r < N) M[(int64_t)r * ld + r] = 1.0;
}

}  // namespace povar
