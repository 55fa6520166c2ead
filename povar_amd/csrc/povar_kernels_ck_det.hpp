// povar_kernels_ck_det.hpp -- the BIT-REPRODUCIBLE form of the camera-chunk E0 operator of step 1 (POVAR_DETERMINISTIC=1;
// a fixed reduction order inside a GPU): right_mul_e0_pOSE (linearization_power_varproj.hpp:364-406) on the
// layout and with the row / record helpers of e0_ck (povar_kernels_ck.hpp, ck_layout.hpp).
//
// e0_ck adds in arrival order in two places: three ds_add_f64 per observation into the landmark's u = Jl^T Jp x, and twelve
// per (camera, tile) into the workgroup's accumulator of y_c -- the GPU's form of the reference's mutex order (:388-398).
// Here neither sum depends on the order the wavefronts arrive in:
//   * u_l is summed in 64-bit FIXED POINT (ds_add_u64: integer addition is associative, every bit of the sum is the same
//     whatever the order).  The binary point of a landmark's sum comes from the data: the way forward is walked TWICE --
//     the first walk leaves the largest exponent of the landmark's contributions (ds_max_i32: order-free as well), the
//     second adds the contributions rounded once to the grid 2^(61 - that exponent - ceil(log2 n_l)) (n_l from the layout):
//     at least 61 - log2 n_l bits below the largest contribution, where fp64 keeps 53 below the running sum.
//     (A-priori bounds were built first -- exponents of max|z|, max|P3|, the image coordinates, |h~| --: the operator norm
//     of the pOSE residual's C(u, v) with pixel coordinates alone is 2^17 above what occurs; the sums kept 25 bits.)
//   * y_c: a chunk's sum is a register sum in row order and a segmented wavefront scan -- fixed already --; the run totals
//     are added into the camera's LDS accumulator in a FIXED ORDER: every (tile, run) has a ticket (CkP::tick, from the
//     layout: batches, then rounds of the tile walk, then the tiles of a round from the shortest to the tallest -- the
//     order they finish in), the accumulator slot carries a counter in LDS, and a run's last lane adds (read, fp64 add,
//     write -- no atomic) when the counter shows its ticket.  A wavefront waits only for tiles that are ahead of its own in
//     every wavefront's program order: no cycle.
// Everything else of the term (cam_cold_sum_binv: strided sums + a butterfly) has a fixed order anyway.
#pragma once

#include "povar_kernels_ck.hpp"
#include "povar_kernels_ck_joint.hpp"

namespace povar {

constexpr int CK_FIX_BITS = 61;
// |v| < 2^ck_xp(v) for v > 0; an exact ZERO contributes no exponent (INT_MIN: a no-op for the maximum) -- frexp's 0 for it
// would set the binary point of a landmark whose real contributions are tiny (late terms: 2^-40) forty bits too high and
// leave the sum ~ 20 bits; inf / nan: 0 (the sum is garbage either way and the finiteness test downstream says so)
__device__ inline int ck_xp(double v) { return v > 0.0 ? __builtin_amdgcn_frexp_exp(v) : (v == 0.0 ? INT_MIN : 0); }
__device__ inline unsigned long long ck_fix(double x, int e) {  // x 2^e to the nearest integer, two's complement (|x 2^e| < 2^62)
  const double xs = __builtin_ldexp(x, e) + 0.5;
  const double hi = __builtin_floor(xs * 0x1p-32);
  const double lo = __builtin_fma(hi, -0x1p32, xs);  // in [0, 2^32]
  const int ih = (int)hi;
  const unsigned il = (unsigned)lo;  // (v_cvt_u32_f64 truncates and saturates: any fixed function of x will do)
  return ((unsigned long long)(unsigned)ih << 32) | il;
}
__device__ inline double ck_unfix(unsigned long long v, int e) {
  const int ih = (int)(unsigned)(v >> 32);
  const unsigned il = (unsigned)v;
  return __builtin_ldexp(__builtin_fma((double)ih, 0x1p32, (double)il), -e);
}
// LDS bytes: e0_ck's arrays + a 16-bit binary point per landmark slot + a 16-bit ticket counter per accumulator slot
__host__ __device__ inline size_t ck_lds_bytes_det(int slots, int n_acc) {
  return ck_lds_bytes_dev(slots, n_acc, 1) + 2 * ((size_t)slots + n_acc) + 16;
}

// one row of the way forward.  PASS 0: the largest exponent of the contributions to the landmark (in the first word of its
// u); PASS 1: the contributions in fixed point
template <int D, bool ROBUST, int PASS>
__device__ inline void ckd_forward_step(const Dp& d, const CkRows& k, CkStream<D, ROBUST>& st, int row0, int li0, int h, int lane,
                                        const double* zz, const double* P3, const double* lh, double* lu, const short* lexp, int j, int i) {
  const double2 uv = st.uv[i];
  const uint32_t s = (st.w[i] >> (16 * (j & 1))) & 0xffffu;
  st.load(k, row0, li0, j + D, h, lane, i);
  if (s != 0xffffu) {
    const double hx = lh[s], hy = lh[s + 1], hz = lh[s + 2];  // (s = 3 x slot)
    const double rw = ROBUST ? ck_huber_w(d, P3, hx, hy, hz, uv) : 1.0;
    LplObs o;
    o.set(d, uv, rw);
    double red[3];
    ck_forward_math(o, zz, P3, hx, hy, hz, red);
    if (PASS == 0) {
      const int e = ck_xp(fmax(fmax(fabs(red[0]), fabs(red[1])), fabs(red[2])));
      __hip_atomic_fetch_max(reinterpret_cast<int*>(lu + s), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      const int e = lexp[(s * 43691u) >> 17];  // (s / 3 for s < 2^16)
#pragma unroll
      for (int m = 0; m < 3; ++m)
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(lu + s + m), ck_fix(red[m], e), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
template <int D, bool ROBUST, int PASS>
__device__ inline void ckd_forward_rows(const Dp& d, const CkRows& k, CkStream<D, ROBUST>& st, int row0, int li0, int h, int lane,
                                        const double* zz, const double* P3, const double* lh, double* lu, const short* lexp) {
  int n0 = 0;
#pragma nounroll
  for (; n0 + D <= h; n0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) ckd_forward_step<D, ROBUST, PASS>(d, k, st, row0, li0, h, lane, zz, P3, lh, lu, lexp, n0 + i, i);
  }
#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (n0 + i < h) ckd_forward_step<D, ROBUST, PASS>(d, k, st, row0, li0, h, lane, zz, P3, lh, lu, lexp, n0 + i, i);
}

// NW wavefronts per workgroup (the count the layout's tiles were scheduled for and its tickets numbered for), SD rows in flight
template <int NW, int SD, bool ROBUST>
__global__ __launch_bounds__(NW * 64) void e0_ck_det(Dp d, CkP k, double* part_out) {
  const int done = d.flags[1];
  extern __shared__ double ck_lds[];
  const CkRows R = ck_rows(k);
  const int S = k.slots;
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* lh = ck_lds + 2;             // [S][3] landmark coordinates of the batch
  double* lu = lh + 3 * S;             // [S][3] u (first the exponent maximum, then fixed point), then g = G u
  double* acc = ck_lds + 2 + 6 * (size_t)S;  // [n_acc][13] per-camera accumulators of the workgroup
  short* lexp = reinterpret_cast<short*>(acc + (size_t)k.max_acc * CK_ACC_STRIDE);  // [S] binary point of u per slot
  short* tick = lexp + S;                                                          // [n_acc] tickets served per accumulator
  const V2& v = d.v2;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_acc = v.wg_cam_off[blockIdx.x + 1] - cam0;
  const int t0 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t1 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  for (int i = threadIdx.x; i < n_acc * CK_ACC_STRIDE; i += NW * 64) acc[i] = 0;
  for (int i = threadIdx.x; i < n_acc; i += NW * 64) tick[i] = 0;
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)k.tile;
  const cint_p bt = (cint_p)(uintptr_t)k.bt_off;
  if (done) return;  // wave-uniform, before any barrier and any side effect
  auto tile_of = [&](int tb0, int q) { return tb0 + q * NW + ((q & 1) ? NW - 1 - wave : wave); };  // (e0_ck's walk without its deal over the SIMDs: the layout's ticket order is cut for this one)
  for (int b = 0; b < k.nb; ++b) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // (per-lane addresses are not carried across the batches: povar_kernels_ck.hpp)
    const int tb0 = bt[blockIdx.x * k.nb + b], tb1 = bt[blockIdx.x * k.nb + b + 1];
    const bool one_tile = tile_of(tb0, 1) >= tb1;  // the wavefront's record stays in registers through the passes
    // ---- landmark coordinates of the batch; the first word of u: the exponent maximum of the first walk
    for (int m = wave; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9) * WAVE + lane;
      const int s = m * WAVE + lane;
      lh[3 * s] = rp[0];
      lh[3 * s + 1] = rp[WAVE];
      lh[3 * s + 2] = rp[2 * WAVE];
      lu[3 * s] = __longlong_as_double((long long)(unsigned)INT_MIN);
      lu[3 * s + 1] = 0;
      lu[3 * s + 2] = 0;
    }
    double zz[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, P3[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    CkStream<SD, ROBUST> st;
    st.clear();
    // first tile of the first walk: in flight across the barrier
    {
      const int t = tile_of(tb0, 0);
      if (t < tb1) {
        const int rank = ck_rank(k.lane_meta[(size_t)t * WAVE + lane].x), rk = rank < 0 ? 0 : rank;
        ck_load_z(d, rk, zz);
        ck_load_p<ROBUST>(d, rk, P3);
        st.template start<1>(R, tiles[4 * t], tiles[4 * t + 3], tiles[4 * t + 1], lane);
      }
    }
    ck_barrier();
    // ---- the way forward, twice
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      for (int q = 0;; ++q) {
        const int t = tile_of(tb0, q);
        if (t >= tb1) break;
        const int row0 = tiles[4 * t], h = tiles[4 * t + 1], li0 = tiles[4 * t + 3];
        if (q > 0 || (pass > 0 && !one_tile)) {
          const int rank = ck_rank(k.lane_meta[(size_t)t * WAVE + lane].x), rk = rank < 0 ? 0 : rank;
          ck_load_z(d, rk, zz);
          ck_load_p<ROBUST>(d, rk, P3);
        }
        if (q > 0 || pass > 0) st.template start<1>(R, row0, li0, h, lane);
        if (pass == 0) ckd_forward_rows<SD, ROBUST, 0>(d, R, st, row0, li0, h, lane, zz, P3, lh, lu, lexp);
        else ckd_forward_rows<SD, ROBUST, 1>(d, R, st, row0, li0, h, lane, zz, P3, lh, lu, lexp);
      }
      if (pass == 0) {
        ck_barrier();
        // the binary point of every landmark's sum: 61 bits - the largest contribution's exponent - ceil(log2(their number))
        for (int m = wave; t0 + b + k.nb * m < t1; m += NW) {
          const int s = m * WAVE + lane;
          const int e = (int)(unsigned)__double_as_longlong(lu[3 * s]);
          const int lc = k.lcnt[(size_t)(t0 + b + k.nb * m) * WAVE + lane];
          lexp[s] = (short)((e == INT_MIN || lc == 255) ? 0 : CK_FIX_BITS - (e + lc));
          lu[3 * s] = 0;
        }
      }
      ck_barrier();
    }
    // ---- g = G u per landmark slot (over u)
    for (int m = wave; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9 + 3) * WAVE + lane;
      const double g0 = rp[0], g1 = rp[WAVE], g2 = rp[2 * WAVE], g3 = rp[3 * WAVE], g4 = rp[4 * WAVE], g5 = rp[5 * WAVE];
      const int s = m * WAVE + lane;
      const int e = lexp[s];
      const double u0 = ck_unfix(__double_as_longlong(lu[3 * s]), e), u1 = ck_unfix(__double_as_longlong(lu[3 * s + 1]), e),
                   u2 = ck_unfix(__double_as_longlong(lu[3 * s + 2]), e);
      lu[3 * s] = g0 * u0 + g1 * u1 + g2 * u2;
      lu[3 * s + 1] = g1 * u0 + g3 * u1 + g4 * u2;
      lu[3 * s + 2] = g2 * u0 + g4 * u1 + g5 * u2;
    }
    // first tile of the way back: P3 (if it has gone), metadata and rows in flight across the barrier
    asm volatile("" : "+v"(lane));
    ck_barrier();
    // ---- the way back: the wavefront's tiles in the order of the walk; run totals into the accumulators in ticket order
    for (int q = 0;; ++q) {
      const int t = tile_of(tb0, q);
      if (t >= tb1) break;
      const int row0 = tiles[4 * t], h = tiles[4 * t + 1], fl = tiles[4 * t + 2], li0 = tiles[4 * t + 3];
      const int2 me = k.lane_meta[(size_t)t * WAVE + lane];
      const int rank = ck_rank(me.x), seg = ck_seg(me.x), acc_slot = me.y;
      const int my_ticket = k.tick[(size_t)t * WAVE + lane];
      if (q > 0 || !one_tile) ck_load_p<ROBUST>(d, rank < 0 ? 0 : rank, P3);
      st.template start<-1>(R, row0, li0, h, lane);
      double y[12];
#pragma unroll
      for (int m = 0; m < 12; ++m) y[m] = 0;
      ck_backward_rows<SD, ROBUST>(d, R, st, row0, li0, h, lane, P3, lh, lu, S, y);
      if (fl & 1) seg_scan_steps<12>(y, lane, seg & 255, 4);  // (inclusive scan: the run's total is in its LAST lane)
      bool pending = false;
      if (rank >= 0) {
        if (acc_slot >= 0) {
          pending = lane == ((seg >> 8) & 255);
        } else {  // a chunk with a partial record of its own
          double2* o = reinterpret_cast<double2*>(part_out + (size_t)(~acc_slot) * 12);
#pragma unroll
          for (int m = 0; m < 6; ++m) o[m] = make_double2(y[2 * m], y[2 * m + 1]);
        }
      }
      const int a_slot = pending ? acc_slot : 0;
      int spins = 0;
      while (__builtin_amdgcn_ballot_w64(pending) != 0) {
        // (every spin is bounded: a layout whose tickets do not match this walk -- which the host never builds -- must end
        // in a reported failure, flags[0] bit 3, not in a hung device)
        const bool give_up = ++spins > (1 << 20);
        if (pending && (give_up || __hip_atomic_load(tick + a_slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (short)my_ticket)) {
          double* a = acc + a_slot * CK_ACC_STRIDE;
#pragma unroll
          for (int m = 0; m < 12; ++m) a[m] += y[m];
          __hip_atomic_store(tick + a_slot, (short)(my_ticket + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (give_up) atomicOr(&d.flags[0], 8);
          pending = false;
        } else if (pending) {
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    ck_barrier();  // the next batch overwrites h~ and u; after the last one: the accumulators are complete
  }
  // ---- accumulators -> this workgroup's partial records (camera-major in part_out)
  for (int i = threadIdx.x; i < n_acc * 6; i += NW * 64) {
    const int r = i / 6, m = 2 * (i % 6);
    const int rec = k.slot_rec[cam0 + r];
    reinterpret_cast<double2*>(part_out + (size_t)rec * 12)[i % 6] = make_double2(acc[r * CK_ACC_STRIDE + m], acc[r * CK_ACC_STRIDE + m + 1]);
  }
}

// ---- step 2: the bit-reproducible form of e0_ck_h (right_mul_e0_poBA, linearization_power_varproj.hpp:408-453), the same
// two measures: the ambient sums U4_l in fixed point (two walks forward), the accumulator adds in ticket order.
__host__ __device__ inline size_t ckh_lds_bytes_det(int n_acc) { return ckh_lds_bytes(n_acc) + 2 * ((size_t)CKH_STRIDE + n_acc) + 16; }

template <int D, bool ROBUST, int PASS>
__device__ inline void ckhd_forward_rows(const CkRows& R, CkStreamH<D, ROBUST>& st, int row0, int li0, int h, int lane,
                                         const Cam& P, const double4 (&zz)[3], const double* lx, double* lu, const short* lexp) {
  auto step = [&](int j, int i) {
    const uint32_t s = (st.w[i] >> (16 * (j & 1))) & 0xffffu;
    const double rw = ROBUST ? st.rw[i] : 1.0;
    st.load(R, row0, li0, j + D, h, lane, i);
    if (s != 0xffffu) {
      const double4 X = make_double4(lx[s], lx[CKH_STRIDE + s], lx[2 * CKH_STRIDE + s], lx[3 * CKH_STRIDE + s]);
      const double sw = ROBUST ? sqrt(rw) : 1.0;
      const Hom hp = ckh_project(P, X);
      double t[2];
      hom_jp_x(hp, X, sw, zz, t);
      const double e0 = sw * hp.D00 * t[0], e1 = sw * hp.D00 * t[1], e2 = sw * (hp.D02 * t[0] + hp.D12 * t[1]);
      const double v[4] = {P.r0.x * e0 + P.r1.x * e1 + P.r2.x * e2, P.r0.y * e0 + P.r1.y * e1 + P.r2.y * e2,
                           P.r0.z * e0 + P.r1.z * e1 + P.r2.z * e2, P.r0.w * e0 + P.r1.w * e1 + P.r2.w * e2};
      if (PASS == 0) {
        const int e = ck_xp(fmax(fmax(fabs(v[0]), fabs(v[1])), fmax(fabs(v[2]), fabs(v[3]))));
        __hip_atomic_fetch_max(reinterpret_cast<int*>(lu + s), e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        const int e = lexp[s];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(lu + m * CKH_STRIDE + s), ck_fix(v[m], e), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
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

template <int NW, int SD, bool ROBUST>
__global__ __launch_bounds__(NW * 64) void e0_ck_h_det(Dp d, CkP k, double* part_out) {
  const int done = d.flags[1];
  extern __shared__ double ck_lds[];
  const CkRows R = ck_rows(k);
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* lx = ck_lds + 2;                  // [4][CKH_STRIDE] X of the batch's landmarks
  double* lu = lx + 4 * CKH_STRIDE;         // [4][CKH_STRIDE] U4 (exponent maximum, then fixed point), then G4
  double* acc = lu + 4 * CKH_STRIDE;        // [n_acc][13]
  short* lexp = reinterpret_cast<short*>(acc + (size_t)k.max_acc * CK_ACC_STRIDE);  // [CKH_STRIDE] binary point of U4 per slot
  short* tick = lexp + CKH_STRIDE;                                                   // [n_acc] tickets served per accumulator
  const V2& v = d.v2;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_acc = v.wg_cam_off[blockIdx.x + 1] - cam0;
  const int t0 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t1 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  for (int i = threadIdx.x; i < n_acc * CK_ACC_STRIDE; i += NW * 64) acc[i] = 0;
  for (int i = threadIdx.x; i < n_acc; i += NW * 64) tick[i] = 0;
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)k.tile;
  const cint_p bt = (cint_p)(uintptr_t)k.bt_off;
  if (done) return;
  auto tile_of = [&](int tb0, int q) { return tb0 + q * NW + ((q & 1) ? NW - 1 - wave : wave); };
  for (int b = 0; b < k.nb; ++b) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tb0 = bt[blockIdx.x * k.nb + b], tb1 = bt[blockIdx.x * k.nb + b + 1];
    const bool one_tile = tile_of(tb0, 1) >= tb1;
    for (int m = wave; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * CKH_REC) * WAVE + lane;
      const int s = m * WAVE + lane;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lx[e * CKH_STRIDE + s] = rp[e * WAVE];
        lu[e * CKH_STRIDE + s] = e == 0 ? __longlong_as_double((long long)(unsigned)INT_MIN) : 0.0;
      }
    }
    double4 zz[3] = {make_double4(0, 0, 0, 0), make_double4(0, 0, 0, 0), make_double4(0, 0, 0, 0)};
    Cam P;
    P.r0 = P.r1 = P.r2 = make_double4(0, 0, 0, 0);
    CkStreamH<SD, ROBUST> st;
    st.clear();
    {
      const int t = tile_of(tb0, 0);
      if (t < tb1) {
        const int rank = ck_rank(k.lane_meta[(size_t)t * WAVE + lane].x);
        ckh_load_rec(d, rank < 0 ? 0 : rank, zz, P);
        st.template start<1>(R, tiles[4 * t], tiles[4 * t + 3], tiles[4 * t + 1], lane);
      }
    }
    ck_barrier();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      for (int q = 0;; ++q) {
        const int t = tile_of(tb0, q);
        if (t >= tb1) break;
        const int row0 = tiles[4 * t], h = tiles[4 * t + 1], li0 = tiles[4 * t + 3];
        if (q > 0 || (pass > 0 && !one_tile)) {
          const int rank = ck_rank(k.lane_meta[(size_t)t * WAVE + lane].x);
          ckh_load_rec(d, rank < 0 ? 0 : rank, zz, P);
        }
        if (q > 0 || pass > 0) st.template start<1>(R, row0, li0, h, lane);
        if (pass == 0) ckhd_forward_rows<SD, ROBUST, 0>(R, st, row0, li0, h, lane, P, zz, lx, lu, lexp);
        else ckhd_forward_rows<SD, ROBUST, 1>(R, st, row0, li0, h, lane, P, zz, lx, lu, lexp);
      }
      if (pass == 0) {
        ck_barrier();
        for (int m = wave; t0 + b + k.nb * m < t1; m += NW) {
          const int s = m * WAVE + lane;
          const int e = (int)(unsigned)__double_as_longlong(lu[s]);
          const int lc = k.lcnt[(size_t)(t0 + b + k.nb * m) * WAVE + lane];
          lexp[s] = (short)((e == INT_MIN || lc == 255) ? 0 : CK_FIX_BITS - (e + lc));
          lu[s] = 0;
        }
      }
      ck_barrier();
    }
    // ---- per landmark slot: U4 (back from fixed point) -> G4
    for (int m = wave; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * CKH_REC + 4) * WAVE + lane;
      double r2[10];
#pragma unroll
      for (int e = 0; e < 10; ++e) r2[e] = rp[e * WAVE];
      const int s = m * WAVE + lane;
      const int ex = lexp[s];
#pragma unroll
      for (int e = 0; e < 4; ++e) lu[e * CKH_STRIDE + s] = ck_unfix(__double_as_longlong(lu[e * CKH_STRIDE + s]), ex);
      ckh_landmark_step(lx, lu, s, r2);
    }
    asm volatile("" : "+v"(lane));
    ck_barrier();
    // ---- the way back: the wavefront's tiles in the order of the walk; run totals into the accumulators in ticket order
    for (int q = 0;; ++q) {
      const int t = tile_of(tb0, q);
      if (t >= tb1) break;
      const int row0 = tiles[4 * t], h = tiles[4 * t + 1], fl = tiles[4 * t + 2], li0 = tiles[4 * t + 3];
      const int2 me = k.lane_meta[(size_t)t * WAVE + lane];
      const int rank = ck_rank(me.x), seg = ck_seg(me.x), acc_slot = me.y;
      const int my_ticket = k.tick[(size_t)t * WAVE + lane];
      if (q > 0 || !one_tile) ckh_load_cam(d, rank < 0 ? 0 : rank, P);
      st.template start<-1>(R, row0, li0, h, lane);
      double y[12];
#pragma unroll
      for (int m = 0; m < 12; ++m) y[m] = 0;
      ckh_backward_rows<SD, ROBUST>(R, st, row0, li0, h, lane, P, lx, lu, y);
      if (fl & 1) seg_scan_steps<12>(y, lane, seg & 255, 4);
      bool pending = false;
      if (rank >= 0) {
        if (acc_slot >= 0) {
          pending = lane == ((seg >> 8) & 255);
        } else {
          double2* o = reinterpret_cast<double2*>(part_out + (size_t)(~acc_slot) * 12);
#pragma unroll
          for (int m = 0; m < 6; ++m) o[m] = make_double2(y[2 * m], y[2 * m + 1]);
        }
      }
      const int a_slot = pending ? acc_slot : 0;
      int spins = 0;
      while (__builtin_amdgcn_ballot_w64(pending) != 0) {
        const bool give_up = ++spins > (1 << 20);  // (bounded: see e0_ck_det)
        if (pending && (give_up || __hip_atomic_load(tick + a_slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (short)my_ticket)) {
          double* a = acc + a_slot * CK_ACC_STRIDE;
#pragma unroll
          for (int m = 0; m < 12; ++m) a[m] += y[m];
          __hip_atomic_store(tick + a_slot, (short)(my_ticket + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (give_up) atomicOr(&d.flags[0], 8);
          pending = false;
        } else if (pending) {
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    ck_barrier();
  }
  for (int i = threadIdx.x; i < n_acc * 6; i += NW * 64) {
    const int r = i / 6, m = 2 * (i % 6);
    const int rc = k.slot_rec[cam0 + r];
    reinterpret_cast<double2*>(part_out + (size_t)rc * 12)[i % 6] = make_double2(acc[r * CK_ACC_STRIDE + m], acc[r * CK_ACC_STRIDE + m + 1]);
  }
}

}  // namespace povar
