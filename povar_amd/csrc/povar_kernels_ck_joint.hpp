// povar_kernels_ck_joint.hpp -- e0_ck_h: the camera-chunk form of the step-2 (RIPOBA, homogeneous landmarks) term kernel.
//
// right_mul_e0_joint (sc/linearization_power_varproj.hpp:408-453), landmark half: the operator of e0_lpl_h
// (povar_kernels_joint.hpp) with the work split as in e0_ck (povar_kernels_ck.hpp): lane = a chunk of observations of ONE
// camera -- its record (z_c 3x4 and P_c 3x4: 24 doubles) and its twelve sums in registers --, the LANDMARKS of a batch in
// LDS.  e0_lpl_h reads eighteen 16-byte record quads per observation and issues twelve LDS atomics (0.27 of the HBM
// peak, LDS-bound); here an observation reads 32 bytes (X) and issues four atomics on the way forward, reads 64 bytes
// (X, g) on the way back.
//
// What makes the split possible is that the landmark's tangent basis N_l and column scale s never have to meet the
// observation: with J4 = sw D P (2 x 4, the ambient landmark Jacobian: hom_jl4 with unit column scale),
//   Jl3 = J4 diag(s) N_l,   u3 = sum Jl3^T t = N_l^T (s .* U4),  U4 = sum J4^T t          (t = Jp12 z_c, 2-vector)
//   Jl3 g3 = J4 (s .* (N_l g3)) = J4 G4
// so the observations accumulate the AMBIENT 4-vector U4 (four atomics) and read the ambient G4 back; N_l (Householder
// vector of X), s and Hll^-1 are applied once per landmark between the passes by the lane that owns the slot.
// The observations' image coordinates are not read at all: the operator depends on P_c and X only (the residual does
// not enter; e0_lpl_h's uv loads are dead code too, which is why its measured bytes are below the "every array once"
// model).  Rows: 2 bytes per observation and pass (+ 8 with a robust norm).
//
// LDS: [8][CKH_STRIDE] doubles (X then U4 / G4, component-major with a COMPILE-TIME stride: every component is an
// immediate offset from one of two addresses; slot -> bank pair is the identity mod 32, the parent layout's row
// placement holds), then the accumulators [n_acc][13].  The layout (ck_layout.hpp: build_ck with slot_bytes = 64,
// li_mul = 1, at most CKH_STRIDE slots) is a second instance beside step 1's.
#pragma once

#include "povar_kernels_ck.hpp"
#include "povar_kernels_joint.hpp"

namespace povar {

// Two strides (round 6): 1536 slots leave room for every accumulator the parent layout has (629 > HOT_ACC_MAX); 2048 slots
// leave 314 -- the layout takes them where they save a landmark batch and keeps, per workgroup, the accumulators of the cameras
// it observes most (ck_layout.hpp: CkShape::wide_slots).  venice-1778 (3 883 landmarks per workgroup): three batches of 1 344
// slots -> two of 2 048, 70.7 -> 64.9 us per term although 314 accumulators instead of 501 turn 87 k more chunks into chunks
// with a record of their own; a batch costs ~ 7 us, the accumulators ~ 2 (profiles/r06_ckh_stride_ab.txt).
constexpr int CKH_STRIDE = 1536;        // landmark slots of a batch at most; component stride of the LDS arrays
constexpr int CKH_STRIDE_WIDE = 2048;   // the same with fewer accumulators beside them (e0_ck_h only; e0_ck_h_det keeps 1536)
constexpr int CKH_REC = 14;        // = LPL_REC_H: doubles per landmark lane in V2::lmrec (step 2): X (4), s (4), Hll^-1 (6)
__host__ __device__ inline size_t ckh_lds_bytes(int n_acc, int stride = CKH_STRIDE) { return 16 + (size_t)8 * stride * 8 + (size_t)n_acc * CK_ACC_STRIDE * 8 + 64; }

// rows of a tile: the landmark-slot words (two 16-bit slots per word) and, with a robust norm, the weights
template <int D, bool ROBUST>
struct CkStreamH {
  uint32_t w[D];
  double rw[D];
  __device__ inline void clear() {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      w[i] = 0xffffffffu;
      rw[i] = 1.0;
    }
  }
  __device__ inline void load(const CkRows& R, int row0, int li0, int j, int h, int lane, int i) {
    j = j < 0 ? 0 : (j >= h ? h - 1 : j);
    const unsigned ul = (unsigned)lane;
    w[i] = __builtin_amdgcn_raw_buffer_load_b32(R.li, ul * 4u, (unsigned)(li0 + (j >> 1)) * (unsigned)(WAVE * 4), 0);
    if (ROBUST) {
      typedef unsigned __attribute__((ext_vector_type(2))) u2;
      const u2 b = __builtin_amdgcn_raw_buffer_load_b64(R.w, ul * 8u, (unsigned)(row0 + j) * (unsigned)(WAVE * 8), 0);
      rw[i] = __longlong_as_double(((long long)b.y << 32) | b.x);
    }
  }
  template <int DIR>
  __device__ inline void start(const CkRows& R, int row0, int li0, int h, int lane) {
#pragma unroll
    for (int i = 0; i < D; ++i) load(R, row0, li0, DIR > 0 ? i : h - 1 - i, h, lane, i);
  }
};

__device__ inline void ckh_load_rec(const Dp& d, int rank, double4 (&zz)[3], Cam& P) {
  const double2* r = reinterpret_cast<const double2*>(d.hot_rec + (size_t)rank * HOT_REC_STRIDE);
  const double2 a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3], a4 = r[4], a5 = r[5];
  zz[0] = make_double4(a0.x, a0.y, a1.x, a1.y);
  zz[1] = make_double4(a2.x, a2.y, a3.x, a3.y);
  zz[2] = make_double4(a4.x, a4.y, a5.x, a5.y);
  const double2 b0 = r[6], b1 = r[7], b2 = r[8], b3 = r[9], b4 = r[10], b5 = r[11];
  P.r0 = make_double4(b0.x, b0.y, b1.x, b1.y);
  P.r1 = make_double4(b2.x, b2.y, b3.x, b3.y);
  P.r2 = make_double4(b4.x, b4.y, b5.x, b5.y);
}
__device__ inline void ckh_load_cam(const Dp& d, int rank, Cam& P) {
  const double2* r = reinterpret_cast<const double2*>(d.hot_rec + (size_t)rank * HOT_REC_STRIDE) + 6;
  const double2 b0 = r[0], b1 = r[1], b2 = r[2], b3 = r[3], b4 = r[4], b5 = r[5];
  P.r0 = make_double4(b0.x, b0.y, b1.x, b1.y);
  P.r1 = make_double4(b2.x, b2.y, b3.x, b3.y);
  P.r2 = make_double4(b4.x, b4.y, b5.x, b5.y);
}

// D of hom_project with ONE division (1 / z; D02 = -x / z^2 as two multiplications: the three correctly rounded divisions of
// hom_project cost thirty instructions per observation and pass; the results differ in the last bit or two)
__device__ inline Hom ckh_project(const Cam& P, const double4& X) {
  const double x = dot4(P.r0, X), y = dot4(P.r1, X), z = dot4(P.r2, X);
  Hom h;
  h.D00 = 1 / z;
  const double iz2 = h.D00 * h.D00;
  h.D02 = -x * iz2;
  h.D12 = -y * iz2;
  h.r0 = h.r1 = 0;
  h.valid = true;
  return h;
}
// J4 = sw D P with D = [D00 0 D02; 0 D00 D12] (hom_jl4 with unit column scale) is never formed:
//   J4^T t = P^T (D^T t),   J4 g = D (P g)       (12 + 4 operations each instead of 16 + 8)
// one observation forward: U4_l += J4^T t,  t = sw D (Z X)
template <bool ROBUST, int CKH_STRIDE = povar::CKH_STRIDE>
__device__ inline void ckh_obs_forward(const Cam& P, const double4 (&zz)[3], double w, const double* lx, double* lu, uint32_t s) {
  const double4 X = make_double4(lx[s], lx[CKH_STRIDE + s], lx[2 * CKH_STRIDE + s], lx[3 * CKH_STRIDE + s]);
  const double sw = ROBUST ? sqrt(w) : 1.0;
  const Hom h = ckh_project(P, X);
  double t[2];
  hom_jp_x(h, X, sw, zz, t);
  const double e0 = sw * h.D00 * t[0], e1 = sw * h.D00 * t[1], e2 = sw * (h.D02 * t[0] + h.D12 * t[1]);
  __hip_atomic_fetch_add(lu + s, P.r0.x * e0 + P.r1.x * e1 + P.r2.x * e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(lu + CKH_STRIDE + s, P.r0.y * e0 + P.r1.y * e1 + P.r2.y * e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(lu + 2 * CKH_STRIDE + s, P.r0.z * e0 + P.r1.z * e1 + P.r2.z * e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(lu + 3 * CKH_STRIDE + s, P.r0.w * e0 + P.r1.w * e1 + P.r2.w * e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// one observation backward: y_c += X (x) q,  q = hom_q(J4 G4)
template <bool ROBUST, int CKH_STRIDE = povar::CKH_STRIDE>
__device__ inline void ckh_obs_backward(const Cam& P, double w, const double* lx, const double* lg, uint32_t s, double (&y)[12]) {
  const double4 X = make_double4(lx[s], lx[CKH_STRIDE + s], lx[2 * CKH_STRIDE + s], lx[3 * CKH_STRIDE + s]);
  const double4 G = make_double4(lg[s], lg[CKH_STRIDE + s], lg[2 * CKH_STRIDE + s], lg[3 * CKH_STRIDE + s]);
  const double sw = ROBUST ? sqrt(w) : 1.0;
  const Hom h = ckh_project(P, X);
  const double p0 = dot4(P.r0, G), p1 = dot4(P.r1, G), p2 = dot4(P.r2, G);
  const double s0 = sw * (h.D00 * p0 + h.D02 * p2);
  const double s1 = sw * (h.D00 * p1 + h.D12 * p2);
  const double4 q = hom_q(h, sw, s0, s1);
  y[0] += X.x * q.x; y[1] += X.y * q.x; y[2] += X.z * q.x; y[3] += X.w * q.x;
  y[4] += X.x * q.y; y[5] += X.y * q.y; y[6] += X.z * q.y; y[7] += X.w * q.y;
  y[8] += X.x * q.z; y[9] += X.y * q.z; y[10] += X.z * q.z; y[11] += X.w * q.z;
}

template <int D, bool ROBUST, int CKH_STRIDE = povar::CKH_STRIDE>
__device__ inline void ckh_forward_rows(const CkRows& R, CkStreamH<D, ROBUST>& st, int row0, int li0, int h, int lane,
                                        const Cam& P, const double4 (&zz)[3], const double* lx, double* lu) {
  auto step = [&](int j, int i) {
    const uint32_t s = (st.w[i] >> (16 * (j & 1))) & 0xffffu;
    const double rw = ROBUST ? st.rw[i] : 1.0;
    st.load(R, row0, li0, j + D, h, lane, i);
    if (s != 0xffffu) ckh_obs_forward<ROBUST, CKH_STRIDE>(P, zz, rw, lx, lu, s);
  };
  int n0 = 0;
#pragma nounroll
  for (; n0 + D <= h; n0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) step(n0 + i, i);
  }
#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (n0 + i < h) step(n0 + i, i);
}
template <int D, bool ROBUST, int CKH_STRIDE = povar::CKH_STRIDE>
__device__ inline void ckh_backward_rows(const CkRows& R, CkStreamH<D, ROBUST>& st, int row0, int li0, int h, int lane,
                                         const Cam& P, const double* lx, const double* lg, double (&y)[12]) {
  auto step = [&](int j, int i) {
    const uint32_t s = (st.w[i] >> (16 * (j & 1))) & 0xffffu;
    const double rw = ROBUST ? st.rw[i] : 1.0;
    st.load(R, row0, li0, j - D, h, lane, i);
    if (s != 0xffffu) ckh_obs_backward<ROBUST, CKH_STRIDE>(P, rw, lx, lg, s, y);
  };
  int n0 = 0;
#pragma nounroll
  for (; n0 + D <= h; n0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) step(h - 1 - (n0 + i), i);
  }
#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (n0 + i < h) step(h - 1 - (n0 + i), i);
}

// Between the passes, per landmark slot (the lane that owns it): U4 -> G4 = s .* (N_l Hll^-1 N_l^T (s .* U4)).
// rec: entries 4..13 of the landmark's record (s (4), Hll^-1 upper triangle (6)); X and U4 are in LDS.
template <int CKH_STRIDE = povar::CKH_STRIDE>
__device__ inline void ckh_landmark_step(double* lx, double* lu, int s, const double (&rec)[10]) {
  const double4 X = make_double4(lx[s], lx[CKH_STRIDE + s], lx[2 * CKH_STRIDE + s], lx[3 * CKH_STRIDE + s]);
  double hw[4], hbeta;
  house4(X, hw, hbeta);
  const double a[4] = {rec[0] * lu[s], rec[1] * lu[CKH_STRIDE + s], rec[2] * lu[2 * CKH_STRIDE + s], rec[3] * lu[3 * CKH_STRIDE + s]};
  const double aw = a[0] * hw[0] + a[1] * hw[1] + a[2] * hw[2] + a[3] * hw[3];
  double u3[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) u3[j] = a[j + 1] - hbeta * aw * hw[j + 1];   // N_l^T a  (jl3_of_jl4 applied to the row a)
  const double* Hi = rec + 4;
  const double g3[3] = {Hi[0] * u3[0] + Hi[1] * u3[1] + Hi[2] * u3[2], Hi[1] * u3[0] + Hi[3] * u3[1] + Hi[4] * u3[2],
                        Hi[2] * u3[0] + Hi[4] * u3[1] + Hi[5] * u3[2]};
  // N_l g3 = [0; g3] - beta (w[1:] . g3) w
  const double gw = hbeta * (hw[1] * g3[0] + hw[2] * g3[1] + hw[3] * g3[2]);
  lu[s] = rec[0] * (-gw * hw[0]);
  lu[CKH_STRIDE + s] = rec[1] * (g3[0] - gw * hw[1]);
  lu[2 * CKH_STRIDE + s] = rec[2] * (g3[1] - gw * hw[2]);
  lu[3 * CKH_STRIDE + s] = rec[3] * (g3[2] - gw * hw[3]);
}

// NW wavefronts per workgroup, SD rows in flight per tile (as e0_ck; one group of wavefronts); CKH_STRIDE: see above -- with the
// wide stride the layout uses the first k.max_acc accumulator slots of the parent layout only (the others' chunks have records)
template <int NW, int SD, bool ROBUST, int CKH_STRIDE = povar::CKH_STRIDE>
__global__ __launch_bounds__(NW * 64) void e0_ck_h(Dp d, CkP k, double* part_out) {
  const int done = d.flags[1];
  extern __shared__ double ck_lds[];
  const CkRows R = ck_rows(k);
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* lx = ck_lds + 2;                  // [4][CKH_STRIDE] X of the batch's landmarks
  double* lu = lx + 4 * CKH_STRIDE;         // [4][CKH_STRIDE] U4, then G4
  double* acc = lu + 4 * CKH_STRIDE;        // [n_acc][13]
  const V2& v = d.v2;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_acc = min(v.wg_cam_off[blockIdx.x + 1] - cam0, k.max_acc);
  const int t0 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t1 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  for (int i = threadIdx.x; i < n_acc * CK_ACC_STRIDE; i += NW * 64) acc[i] = 0;
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)k.tile;
  const cint_p bt = (cint_p)(uintptr_t)k.bt_off;
  if (done) return;  // wave-uniform, before any barrier and any side effect
  auto tile_of = [&](int tb0, int q) { return tb0 + q * NW + ((q & 1) ? NW - 1 - wave : wave); };
  constexpr int HM = 32 / NW > 0 ? 32 / NW : 1;  // slot tiles per wavefront whose X is requested a phase ahead
  double xn[HM][4];
  auto request_x = [&](int b, int lane) {
#pragma unroll
    for (int q = 0; q < HM; ++q) {
      xn[q][0] = xn[q][1] = xn[q][2] = 0;
      xn[q][3] = 1;
      const int m = wave + q * NW;
      if (b < k.nb && t0 + b + k.nb * m < t1) {
        const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * CKH_REC) * WAVE + lane;
        xn[q][0] = rp[0]; xn[q][1] = rp[WAVE]; xn[q][2] = rp[2 * WAVE]; xn[q][3] = rp[3 * WAVE];
      }
    }
  };
  // (the batches' tile ranges from LDS after the first batch, the next batch's first requests behind the barrier that ends
  // the way forward: as e0_ck -- the lgkmcnt(0) in front of a barrier waits for every scalar load in flight)
  int* lbt = reinterpret_cast<int*>(acc + (size_t)k.max_acc * CK_ACC_STRIDE);
  const bool bt_lds = k.nb + 1 <= 16;
  if (bt_lds && (int)threadIdx.x <= k.nb) lbt[threadIdx.x] = k.bt_off[blockIdx.x * k.nb + threadIdx.x];
  auto bt_of = [&](int i, bool first) {
    return (bt_lds && !first) ? __builtin_amdgcn_readfirstlane(lbt[i]) : bt[blockIdx.x * k.nb + i];
  };
  int rank_next = 0;
  auto request_first_meta = [&](int b, int lane) {
    rank_next = 0;
    if (b < k.nb) {
      const int tb0 = bt_of(b, b == 0), tb1 = bt_of(b + 1, b == 0);
      if (tb0 + wave < tb1) rank_next = ck_rank(k.lane_meta[(size_t)(tb0 + wave) * WAVE + lane].x);
    }
  };
  request_first_meta(0, lane0);
  request_x(0, lane0);
  for (int b = 0; b < k.nb; ++b) {
    // (opaque lane number per batch and pass, every array assigned on every path: see e0_ck)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tb0 = bt_of(b, b == 0), tb1 = bt_of(b + 1, b == 0);
    int q_t = 0;
    int t = tile_of(tb0, 0);
    int rank = rank_next;
    double4 zz[3] = {make_double4(0, 0, 0, 0), make_double4(0, 0, 0, 0), make_double4(0, 0, 0, 0)};
    Cam P;
    P.r0 = P.r1 = P.r2 = make_double4(0, 0, 0, 0);
    CkStreamH<SD, ROBUST> st;
    st.clear();
    int row0 = 0, h = 0, fl = 0, li0 = 0;
    int tn = tile_of(tb0, 1);
    int rank_n = 0;
    // ---- the way forward starts: record and first rows of the first tile
    if (t < tb1) {
      row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];  // (the whole header at once: below)
      if (tn < tb1) rank_n = ck_rank(k.lane_meta[(size_t)tn * WAVE + lane].x);
      ckh_load_rec(d, rank < 0 ? 0 : rank, zz, P);
      st.template start<1>(R, row0, li0, h, lane);
    }
    // ---- X of the batch into LDS (requested a phase ago), U4 = 0
#pragma unroll
    for (int q = 0; q < HM; ++q) {
      const int m = wave + q * NW;
      if (t0 + b + k.nb * m < t1) {
        const int s = m * WAVE + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lx[e * CKH_STRIDE + s] = xn[q][e];
          lu[e * CKH_STRIDE + s] = 0;
        }
      }
    }
    for (int m = wave + HM * NW; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * CKH_REC) * WAVE + lane;
      const int s = m * WAVE + lane;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lx[e * CKH_STRIDE + s] = rp[e * WAVE];
        lu[e * CKH_STRIDE + s] = 0;
      }
    }
    ck_barrier();
    // ---- forward
    while (t < tb1) {
      ckh_forward_rows<SD, ROBUST, CKH_STRIDE>(R, st, row0, li0, h, lane, P, zz, lx, lu);
      if (tn >= tb1) break;  // (t, q_t, rank, P stay on the last tile: the way back starts there)
      t = tn;
      ++q_t;
      rank = rank_n;
      tn = tile_of(tb0, q_t + 1);
      row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];
      if (tn < tb1) rank_n = ck_rank(k.lane_meta[(size_t)tn * WAVE + lane].x);
      ckh_load_rec(d, rank < 0 ? 0 : rank, zz, P);
      st.template start<1>(R, row0, li0, h, lane);
    }
    // ---- the way back starts before the barriers in front of it
    asm volatile("" : "+v"(lane));
    int acc_slot = 0, seg = 0;
    int tp = q_t > 0 ? tile_of(tb0, q_t - 1) : tb1;
    int rank_p = 0, acc_p = 0, seg_p = 0;
    double rec[HM][10];  // s (4) and Hll^-1 (6) of the wavefront's landmark slots
#pragma unroll
    for (int q = 0; q < HM; ++q) {
#pragma unroll
      for (int e = 0; e < 10; ++e) rec[q][e] = 0;
      const int m = wave + q * NW;
      if (t0 + b + k.nb * m < t1) {
        const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * CKH_REC + 4) * WAVE + lane;
#pragma unroll
        for (int e = 0; e < 10; ++e) rec[q][e] = rp[e * WAVE];
      }
    }
    if (t < tb1) {
      const int2 me = k.lane_meta[(size_t)t * WAVE + lane];
      seg = ck_seg(me.x);
      acc_slot = me.y;  // (fl came with the tile's header: a scalar load HERE is a miss in front of the barrier's lgkmcnt(0) --
      if (tp < tb1) {   //  ~ 2.6 k cycles per batch in e0_ck's stamps, profiles/r06_e0_ck_phase_stamps.txt)
        const int2 mp = k.lane_meta[(size_t)tp * WAVE + lane];
        rank_p = ck_rank(mp.x);
        seg_p = ck_seg(mp.x);
        acc_p = mp.y;
      }
      st.template start<-1>(R, row0, li0, h, lane);
    }
    ck_barrier();
    request_first_meta(b + 1, lane);
    request_x(b + 1, lane);
    // ---- per landmark slot: U4 -> G4
#pragma unroll
    for (int q = 0; q < HM; ++q) {
      const int m = wave + q * NW;
      if (t0 + b + k.nb * m < t1) ckh_landmark_step<CKH_STRIDE>(lx, lu, m * WAVE + lane, rec[q]);
    }
    for (int m = wave + HM * NW; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * CKH_REC + 4) * WAVE + lane;
      double r2[10];
#pragma unroll
      for (int e = 0; e < 10; ++e) r2[e] = rp[e * WAVE];
      ckh_landmark_step<CKH_STRIDE>(lx, lu, m * WAVE + lane, r2);
    }
    ck_barrier();
    // ---- backward: the wavefront's tiles in reverse
    while (t < tb1) {
      double y[12];
#pragma unroll
      for (int m = 0; m < 12; ++m) y[m] = 0;
      ckh_backward_rows<SD, ROBUST, CKH_STRIDE>(R, st, row0, li0, h, lane, P, lx, lu, y);
      ck_flush_tile(y, fl, lane, rank, acc_slot, seg, acc, n_acc, part_out);
      if (tp >= tb1) break;
      t = tp;
      --q_t;
      rank = rank_p; acc_slot = acc_p; seg = seg_p;
      tp = q_t > 0 ? tile_of(tb0, q_t - 1) : tb1;
      row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];
      if (tp < tb1) {
        const int2 mp = k.lane_meta[(size_t)tp * WAVE + lane];
        rank_p = ck_rank(mp.x);
        seg_p = ck_seg(mp.x);
        acc_p = mp.y;
      }
      ckh_load_cam(d, rank < 0 ? 0 : rank, P);
      st.template start<-1>(R, row0, li0, h, lane);
    }
    ck_barrier();  // the next batch overwrites X and U4; after the last one: the accumulators are complete
  }
  // ---- accumulators -> this workgroup's partial records (camera-major in part_out)
  const __amdgpu_buffer_rsrc_t PR = ck_part_rsrc(part_out);
  for (int i = threadIdx.x; i < n_acc * 6; i += NW * 64) {
    const int r = i / 6, m = 2 * (i % 6);
    const int rc = k.slot_rec[cam0 + r];
    ck_store_part(PR, (unsigned)rc * 96u + 16u * (unsigned)(i % 6), acc[r * CK_ACC_STRIDE + m], acc[r * CK_ACC_STRIDE + m + 1]);
  }
  if (d.p2p_epoch && blockIdx.x == 0 && threadIdx.x == 0) *d.p2p_epoch += 1;  // one tick per term (as e0_lpl_h)
}

}  // namespace povar
