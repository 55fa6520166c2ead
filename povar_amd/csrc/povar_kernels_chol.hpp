// povar_kernels_chol.hpp -- dense fp64 Cholesky solve of the reduced camera system for
// --solver-type-step-1 CHOLESKY (solve_direct_pOSE, sc/linearization_sc.hpp:236-245, where the
// reference calls Eigen::SimplicialLLT on the assembled sparse S).
//
// Layout: M is row-major N x LD, N = 12 n_cams rounded up to 64, LD = N + 64.  Columns [0, N) hold the
// upper triangle of S (identity on the padding rows), column N holds the right-hand side -b, the rest
// of the last 64-column tile is zero.  Factorisation S = R^T R (R upper, row-major), right-looking
// in panels of 64 rows:
//   chol_diag   R11 = chol(S11)                           one workgroup, block in LDS
//   chol_trsm   R12 = R11^-T S12 (and y_k = R11^-T rhs_k)   one thread per column
//   chol_syrk   S22 -= R12^T R12 (and rhs -= R12^T y_k)     64x64 tiles, v_mfma_f64_16x16x4_f64
// Carrying the right-hand side as column N folds the forward substitution into the factorisation;
// chol_back then solves R x = y panel by panel from the bottom.
// The trailing update is the n^3/3 part (venice-1778: 3.2 TFLOP): the only GEMM-shaped work of
// the whole path, so it runs on the fp64 matrix cores; both operands of R12^T R12 are rows of the
// same row panel, so the A and B fragments are plain coalesced row reads staged in LDS.
#pragma once
#include "povar_kernels.hpp"

namespace povar {

constexpr int CH_NB = 64;    // panel height == tile edge
constexpr int CH_LDS = 80;   // LDS row stride in doubles: rows 2 apart share banks, the 32-lane halves of a b64 read do not

typedef double ch_d4 __attribute__((ext_vector_type(4)));

// R11 = chol(S11): upper Cholesky of the 64 x 64 diagonal block at (k0, k0); info |= 1 when a pivot
// is not positive
__global__ __launch_bounds__(256) void chol_diag(double* M, int64_t ld, int k0, int* info) {
  __shared__ double A[CH_NB][CH_NB + 1];
  double* blk = M + (int64_t)k0 * ld + k0;
  for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 256) A[e >> 6][e & 63] = blk[(int64_t)(e >> 6) * ld + (e & 63)];
  __syncthreads();
  for (int j = 0; j < CH_NB; ++j) {
    const double piv = A[j][j];
    if (threadIdx.x == 0 && !(piv > 0)) atomicOr(info, 1);
    const double d = sqrt(piv);
    __syncthreads();
    for (int c = j + threadIdx.x; c < CH_NB; c += 256) A[j][c] = (c == j) ? d : A[j][c] / d;
    __syncthreads();
    // trailing update of the upper triangle: A[r][c] -= R[j][r] R[j][c], j < r <= c
    const int m = CH_NB - 1 - j;
    for (int e = threadIdx.x; e < m * m; e += 256) {
      const int r = j + 1 + e / m, c = j + 1 + e % m;
      if (c >= r) A[r][c] -= A[j][r] * A[j][c];
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    blk[(int64_t)r * ld + c] = c >= r ? A[r][c] : 0.0;
  }
}

// R12 = R11^-T S12 for the columns [k0 + 64, LD): forward substitution down each column,
// R11 broadcast from LDS
__global__ __launch_bounds__(128) void chol_trsm(double* M, int64_t ld, int k0) {
  __shared__ double R[CH_NB][CH_NB];
  const double* blk = M + (int64_t)k0 * ld + k0;
  for (int e = threadIdx.x; e < CH_NB * CH_NB; e += 128) R[e >> 6][e & 63] = blk[(int64_t)(e >> 6) * ld + (e & 63)];
  __syncthreads();
  const int64_t c = (int64_t)k0 + CH_NB + (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (c >= ld) return;
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

// S22 -= R12^T R12 on the upper block triangle of the trailing matrix (including the rhs tile).
// grid = (n_jt, n_it); tile (it, jt) with jt >= it covers rows k1 + 64 it, columns k1 + 64 jt.
// 4 wavefronts per workgroup, each a 32 x 32 quadrant = 2 x 2 MFMA tiles; fragments:
// A (16x4): lane l holds panel[kk + (l >> 4)][i + (l & 15)], B likewise with j, D: col = l & 15,
// row = (l >> 4) + 4 reg.
__global__ __launch_bounds__(256) void chol_syrk(double* M, int64_t ld, int k0) {
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

// back substitution R x = y for the panel at k0 (called from the last panel to the first):
// t_i = y_i - sum_{c >= k0 + 64} R[k0 + i][c] x[c], then the 64 x 64 triangular solve in one wavefront.
// y lives in column N of M; x is a separate vector of length N.
__global__ __launch_bounds__(1024) void chol_back(const double* M, int64_t ld, int N, int k0, double* x) {
  __shared__ double t[CH_NB];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;  // 16 wavefronts, 4 rows each
  for (int i = w; i < CH_NB; i += 16) {
    const double* row = M + (int64_t)(k0 + i) * ld;
    double s[1] = {0};
    for (int c = k0 + CH_NB + lane; c < N; c += WAVE) s[0] += row[c] * x[c];
    wave_sum<1>(s);
    if (lane == 0) t[i] = row[N] - s[0];
  }
  __syncthreads();
  if (w != 0) return;
  double xi = 0;  // lane p holds x[k0 + p]
  for (int i = CH_NB - 1; i >= 0; --i) {
    const double* row = M + (int64_t)(k0 + i) * ld + k0;
    double s[1] = {lane > i ? row[lane] * xi : 0.0};
    wave_sum<1>(s);
    const double v = (t[i] - s[0]) / row[i];
    if (lane == i) xi = v;
  }
  x[k0 + lane] = xi;
}

// identity on the padding rows [n, N) of the augmented matrix (after the memset)
__global__ __launch_bounds__(64) void chol_pad(double* M, int64_t ld, int n, int N) {
  const int r = n + threadIdx.x;
  if (r < N) M[(int64_t)r * ld + r] = 1.0;
}

}  // namespace povar
