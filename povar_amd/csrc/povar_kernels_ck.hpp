// povar_kernels_ck.hpp -- gfx950 device code of the camera-chunk form of the per-term E0 operator (step 1):
// right_mul_e0_pOSE (linearization_power_varproj.hpp:364-406) with lane = a chunk of one camera's observations.
//
// e0_lpl (povar_kernels.hpp) keeps the CAMERAS of a workgroup in LDS and a landmark in every lane: per observation it
// reads 240 bytes of camera record from LDS and issues twelve ds_add_f64 -- the LDS pipe is 73 % busy and bounds the
// kernel (profiles/r03_pmc_sq.csv).  e0_ck turns the split round:
//   * lane = a chunk of <= CK_HMAX observations of ONE camera inside a landmark batch: the camera's record (Z = sigma_c x_c
//     as 3x4, P3 = P_c[:, :3]) is gathered once per chunk from the rank-ordered structure-of-arrays image (L2 hits, a
//     coalesced 8-byte load per entry when the lanes hold neighbouring ranks) and lives in REGISTERS, and so does the
//     chunk's accumulator y_c (12 doubles): the reference's mutex-guarded `res += Jp^T s` (:388-398) becomes a
//     register sum, one segmented wavefront sum per tile and one LDS add per (camera, tile);
//   * the LANDMARKS of the batch live in LDS: h~ (24 bytes) and u = Jl^T Jp x, later g = G u (24 bytes).  Forward, per
//     observation: three ds_read_b64 + three ds_add_f64; backward: six ds_read_b64, no atomics.
// A batch = the lane-per-landmark tiles {b, b + NB, ...} of the workgroup (ck_layout.hpp), so V2::lmrec -- written by
// prepare_lpl in lane order -- is read as it stands.  Phases per batch, separated by workgroup barriers:
//   load h~ -> forward (all chunk tiles) -> g = G u per landmark slot -> backward (all chunk tiles).
// Every row (18 bytes per observation) is read on both passes.  With the HUBER norm the observation's weight is NOT a third
// row array (8 bytes per observation and pass more: round 4): the lane recomputes it from what it holds anyway -- the camera
// at the linearisation point (P3 and the translation column, registers), the landmark (LDS) and the image point (ck_huber_w).  (Keeping the rows of a batch in registers between the
// passes was built and measured: at the four wavefronts per SIMD the kernel needs for latency there are 128 VGPRs per
// lane, the forward pass alone needs ~120 of them, and with 8 or 12 wavefronts per workgroup it ran twice as long.)
#pragma once

#include "povar_kernels.hpp"

namespace povar {

constexpr int CK_ROWS = 16;  // = CK_HMAX of ck_layout.hpp: rows per chunk tile at most

struct CkP {
  const double2* uv;     // [rows][64] image points; or, packed (ck_layout.hpp: ck_pack_uv), int2 (k_u, k_v) with u = RN(k_u / 10^6)
  const uint32_t* li;    // [li_rows][64] two 16-bit words per entry: 3 x the landmark's slot (its first LDS double), 0xffff: none
  const double* w;       // [rows][64] robust weights (only with a robust norm)
  const int4* tile;      // first row, height, flags, first li row
  const int2* lane_meta; // [tiles][64] x: rank of the lane's camera | first << 16 | last << 22 lane sharing its accumulator
                         // (x < 0: empty lane); y: >= 0 accumulator slot, < 0: ~(partial record of a cold chunk)
  const int* bt_off;     // [grid * nb + 1]
  const int* slot_rec;   // partial record of each workgroup slot
  int nb, slots;
  unsigned uv_bytes, li_bytes;  // sizes of uv (16 bytes per entry, 8 packed; w's: 8) and li: the rows are read through buffer descriptors
  // the bit-reproducible form (e0_ck_det, povar_kernels_ck_det.hpp)
  const uint8_t* lcnt;     // [lpl tiles][64] ceil(log2(observations added into the lane's landmark slot)); 255: none
  const uint16_t* tick;    // [tiles][64] ticket of the lane's run total at its accumulator slot (last lane of a run with a slot)
  int max_acc;             // accumulator slots the LDS is laid out for
  int uv_packed;           // uv holds packed image points (the PK instantiations of e0_ck)
  const int* cpos;         // [rows][64] cold-view position of the entries of chunks WITHOUT an accumulator slot (CkLayout::cpos), or nullptr:
  double4* q4c;            // ... such a lane stores q of every observation there (the per-camera kernel forms h~ (x) q), no record
};
// a packed image coordinate back to the double it was packed from: the sequence ck_pack_uv (ck_layout.hpp) verified on the host
// for every entry -- int -> double, a multiplication, two fused multiply-adds: correctly rounded operations, the same bits
__device__ inline double ck_unpack_uv(unsigned k) {
  const double kd = (double)(int)k;
  const double q0 = kd * 1e-6;
  const double r = __builtin_fma(-q0, 1e6, kd);
  return __builtin_fma(r, 1e-6, q0);
}
typedef unsigned __attribute__((ext_vector_type(4))) ck_u4;
constexpr int CK_ACC_STRIDE = 13;  // doubles per accumulator slot in LDS (12 used)
__host__ __device__ inline size_t ck_lds_bytes_dev(int slots, int n_acc, int ng) { return 16 + (size_t)ng * slots * 48 + (size_t)n_acc * CK_ACC_STRIDE * 8 + 64; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release + acquire fence around s_barrier and
// waits for EVERY outstanding vector memory operation (s_waitcnt vmcnt(0)) first -- including the loads this kernel
// issues ahead of its barriers precisely so that they are in flight while the wavefront waits.  What the phases hand
// over through the barriers is in LDS (h~, u, g, the accumulators): lgkmcnt(0) is all the ordering they need; the
// global loads stay in flight and the compiler waits for each where its registers are first used.
__device__ inline void ck_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Barrier of ONE group of wavefronts of the workgroup (NG > 1: the workgroup's wavefronts work as NG independent groups on
// different landmark batches, so that one group's latency phases -- record gathers, second tiles, barrier waits --
// overlap the other's fp64-bound row phases on the same SIMDs).  s_barrier is workgroup-wide, so this one is a counter
// in LDS: every wavefront adds one (after its own LDS traffic has drained) and sleeps until the group's count reaches
// its generation.  LDS is one coherent memory for the CU: what the other wavefronts wrote before their add is visible.
__device__ inline void ck_group_barrier(int* cnt, int& gen, int group_waves, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  gen += group_waves;
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) __builtin_amdgcn_s_sleep(2);
  asm volatile("" ::: "memory");
}

// lane metadata word: rank | first << 16 | last << 22 (negative: empty lane); seg comes out as first | last << 8
__host__ __device__ inline int ck_rank(int x) { return x < 0 ? -1 : (x & 0xffff); }
__host__ __device__ inline int ck_seg(int x) { return ((x >> 16) & 63) | (((x >> 22) & 63) << 8); }

// lpl_forward without the accumulation (red = ..., not red += ...: three fp64 adds per observation less)
__device__ inline void ck_forward_math(const LplObs& o, const double* zz, const double* P3, double hx, double hy, double hz,
                                       double* red) {
  const double d0 = hx * zz[0] + hy * zz[1] + hz * zz[2] + zz[3];
  const double d1 = hx * zz[4] + hy * zz[5] + hz * zz[6] + zz[7];
  const double d2 = hx * zz[8] + hy * zz[9] + hz * zz[10] + zz[11];
  const double a0 = o.w * (d0 - o.cu * d2);
  const double a1 = o.w * (d1 - o.cv * d2);
  const double a2 = o.w * (o.cuv * d2 - o.cu * d0 - o.cv * d1);
  red[0] = P3[0] * a0 + P3[3] * a1 + P3[6] * a2;
  red[1] = P3[1] * a0 + P3[4] * a1 + P3[7] * a2;
  red[2] = P3[2] * a0 + P3[5] * a1 + P3[8] * a2;
}

// The HUBER weight of an observation at the linearisation point (compute_error_weight, bal_bundle_adjustment_helper.cpp:52-74,
// on the pOSE residual :250-261), from the camera P = [P3 | t] (P3[0..8] row-major, P3[9..11] = t), the landmark and the
// image point: with p = P3 h + t the residual is (sb (p0 - u p2), sb (p1 - v p2), sa (p0 - u), sa (p1 - v)).
// (The lane-per-landmark linearisation stores the same number in V2::w; the two differ by rounding only, and the weight
// is continuous where the norm switches branches.)
__device__ inline double ck_huber_w(const Dp& d, const double* P3, double hx, double hy, double hz, double2 uv) {
  const double p0 = P3[0] * hx + P3[1] * hy + P3[2] * hz + P3[9];
  const double p1 = P3[3] * hx + P3[4] * hy + P3[5] * hz + P3[10];
  const double p2 = P3[6] * hx + P3[7] * hy + P3[8] * hz + P3[11];
  const double a = p0 - uv.x * p2, b = p1 - uv.y * p2, c = p0 - uv.x, e = p1 - uv.y;
  const double r2 = d.sb * d.sb * (a * a + b * b) + d.sa * d.sa * (c * c + e * e);
  // w = min(1, t / sqrt(r2)) without the division and the square root (some sixty instructions whenever one lane of the
  // wavefront has an outlier): the hardware's reciprocal square root (about 2^-26) and two Newton steps, every lane
  const double t = d.huber;
  double y = __builtin_amdgcn_rsq(r2);
  const double hr = 0.5 * r2;
  y = y * __builtin_fma(-hr * y, y, 1.5);
  y = y * __builtin_fma(-hr * y, y, 1.5);
  return r2 < t * t ? 1.0 : t * y;
}

// One observation forward: u_l += P3^T (w C (Z h~_l)); backward: y_c += h~_l (x) (w C (P3 g_l))
__device__ inline void ck_obs_forward(const Dp& d, double2 uv, double w, const double* zz, const double* P3, double hx, double hy,
                                      double hz, double* lu, int S, uint32_t s) {
  LplObs o;
  o.set(d, uv, w);
  double red[3];
  ck_forward_math(o, zz, P3, hx, hy, hz, red);
  __hip_atomic_fetch_add(lu + s, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (s = 3 x slot)
  __hip_atomic_fetch_add(lu + s + 1, red[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(lu + s + 2, red[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void ck_obs_backward(const Dp& d, double2 uv, double w, const double* P3, double hx, double hy, double hz,
                                       const double* g, double* y) {
  LplObs o;
  o.set(d, uv, w);
  double q[3];
  lpl_backward(o, P3, g, q);
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    y[4 * m] += hx * q[m];
    y[4 * m + 1] += hy * q[m];
    y[4 * m + 2] += hz * q[m];
    y[4 * m + 3] += q[m];
  }
}

// Streamed tile (rows read where they are used): D rows are in flight ahead of the row being worked on.  The D row
// buffers are statically indexed -- the loop over the rows is unrolled by D -- so that a row's loads really have D
// iterations to land.  (A rolled loop with a shift register of D buffers was built first: the register moves of
// iteration j + 1 touch what iteration j has just requested, so every row waited for its predecessor's loads whatever
// D was -- s_waitcnt vmcnt(0) at the top of the loop, 1400 cycles per row on the way back.)  D is even: the two
// landmark slots of an li word then sit at a static shift.
// buffer descriptors of the row arrays (wave-uniform: built once per kernel from kernel arguments)
struct CkRows {
  __amdgpu_buffer_rsrc_t uv, li, w;  // (w: step 2's e0_ck_h only)
};
__device__ inline CkRows ck_rows(const CkP& k) {
  CkRows R;
  R.uv = __builtin_amdgcn_make_buffer_rsrc(const_cast<double2*>(k.uv), 0, k.uv_bytes, 0x00020000);
  R.li = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(k.li), 0, k.li_bytes, 0x00020000);
  R.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(k.w), 0, k.w ? (k.uv_packed ? k.uv_bytes : k.uv_bytes / 2) : 0, 0x00020000);
  return R;
}
template <int D, bool ROBUST, bool PK = false>
struct CkStream {
  double2 uv[D];
  uint32_t w[D];
  __device__ inline void clear() {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      uv[i] = make_double2(0, 0);
      w[i] = 0xffffffffu;
    }
  }
  // buffer i <- row j of the tile (j clamped into the tile: a request past its end re-reads its last row -- a cache hit --
  // so that every step issues the same loads and the wait counters can be exact: with loads under `if (j < h)` the
  // compiler waited for all but the newest load, i.e. for the row it had requested one step earlier).
  // The rows are read through buffer descriptors (CkRows): descriptor in SGPRs + the lane's constant 32-bit byte offset +
  // the row's byte offset as the scalar offset -- no VALU instruction per load (a flat load took a 64-bit add each, two
  // per row and pass in loops that are VALU-bound).
  __device__ inline void load(const CkRows& R, int row0, int li0, int j, int h, int lane, int i) {
    j = j < 0 ? 0 : (j >= h ? h - 1 : j);
    const unsigned ul = (unsigned)lane;
    const unsigned ro = (unsigned)(row0 + j) * (unsigned)(WAVE * 16), lo = (unsigned)(li0 + (j >> 1)) * (unsigned)(WAVE * 4);
    if (PK) {  // packed image points: 8 bytes per observation
      typedef unsigned __attribute__((ext_vector_type(2))) u2;
      const u2 a = __builtin_amdgcn_raw_buffer_load_b64(R.uv, ul * 8u, ro >> 1, 0);
      uv[i] = make_double2(ck_unpack_uv(a.x), ck_unpack_uv(a.y));
    } else {
      typedef unsigned __attribute__((ext_vector_type(4))) u4;
      const u4 a = __builtin_amdgcn_raw_buffer_load_b128(R.uv, ul * 16u, ro, 0);
      uv[i] = make_double2(__longlong_as_double(((long long)a.y << 32) | a.x), __longlong_as_double(((long long)a.w << 32) | a.z));
    }
    w[i] = __builtin_amdgcn_raw_buffer_load_b32(R.li, ul * 4u, lo, 0);
  }
  // step n of the walk is row n (DIR = +1) or row h - 1 - n (DIR = -1: the way back starts with the rows the way forward
  // read last, the ones most likely still in the XCD's L2); buffer n % D holds it
  template <int DIR>
  __device__ inline void start(const CkRows& R, int row0, int li0, int h, int lane) {
#pragma unroll
    for (int i = 0; i < D; ++i) load(R, row0, li0, DIR > 0 ? i : h - 1 - i, h, lane, i);
  }
};
// the rows of one tile (h >= 1); st has been started on the tile (steps 0 .. D-1 are in flight)
template <int D, bool ROBUST, bool PK>
__device__ inline void ck_forward_step(const Dp& d, const CkRows& k, CkStream<D, ROBUST, PK>& st, int row0, int li0, int h, int lane,
                                       const double* zz, const double* P3, const double* lh, double* lu, int S, int j, int i) {
  const double2 uv = st.uv[i];
  const uint32_t s = (st.w[i] >> (16 * (j & 1))) & 0xffffu;
  st.load(k, row0, li0, j + D, h, lane, i);
  if (s != 0xffffu) {
    const double hx = lh[s], hy = lh[s + 1], hz = lh[s + 2];  // (s = 3 x slot: ck_layout.hpp)
    const double rw = ROBUST ? ck_huber_w(d, P3, hx, hy, hz, uv) : 1.0;
    ck_obs_forward(d, uv, rw, zz, P3, hx, hy, hz, lu, S, s);
  }
}
template <int D, bool ROBUST, bool PK>
__device__ inline void ck_forward_rows(const Dp& d, const CkRows& k, CkStream<D, ROBUST, PK>& st, int row0, int li0, int h, int lane,
                                       const double* zz, const double* P3, const double* lh, double* lu, int S) {
  int n0 = 0;
#pragma nounroll
  for (; n0 + D <= h; n0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) ck_forward_step<D, ROBUST, PK>(d, k, st, row0, li0, h, lane, zz, P3, lh, lu, S, n0 + i, i);
  }
#pragma unroll
  for (int i = 0; i < D - 1; ++i)  // the last h % D rows
    if (n0 + i < h) ck_forward_step<D, ROBUST, PK>(d, k, st, row0, li0, h, lane, zz, P3, lh, lu, S, n0 + i, i);
}
template <int D, bool ROBUST, bool PK>
__device__ inline void ck_backward_step(const Dp& d, const CkRows& k, CkStream<D, ROBUST, PK>& st, int row0, int li0, int h, int lane,
                                        const double* P3, const double* lh, const double* lg, int S, double* y, int j, int i) {
  const double2 uv = st.uv[i];
  const uint32_t s = (st.w[i] >> (16 * (j & 1))) & 0xffffu;
  st.load(k, row0, li0, j - D, h, lane, i);
  if (s != 0xffffu) {
    const double hx = lh[s], hy = lh[s + 1], hz = lh[s + 2];
    const double rw = ROBUST ? ck_huber_w(d, P3, hx, hy, hz, uv) : 1.0;
    const double g[3] = {lg[s], lg[s + 1], lg[s + 2]};
    ck_obs_backward(d, uv, rw, P3, hx, hy, hz, g, y);
  }
}
template <int D, bool ROBUST, bool PK>
__device__ inline void ck_backward_rows(const Dp& d, const CkRows& k, CkStream<D, ROBUST, PK>& st, int row0, int li0, int h, int lane,
                                        const double* P3, const double* lh, const double* lg, int S, double* y) {
  int n0 = 0;
#pragma nounroll
  for (; n0 + D <= h; n0 += D) {
#pragma unroll
    for (int i = 0; i < D; ++i) ck_backward_step<D, ROBUST, PK>(d, k, st, row0, li0, h, lane, P3, lh, lg, S, y, h - 1 - (n0 + i), i);
  }
#pragma unroll
  for (int i = 0; i < D - 1; ++i)
    if (n0 + i < h) ck_backward_step<D, ROBUST, PK>(d, k, st, row0, li0, h, lane, P3, lh, lg, S, y, h - 1 - (n0 + i), i);
}

// The way back over a tile that has lanes of cameras WITHOUT an accumulator slot, where those lanes leave q per observation in
// the cold view (CkP::cpos / q4c) instead of a record of their own: such tiles are the last ones of a batch (chunks are sorted
// by length, cold chunks are almost all of one observation: one or two rows), so the rows are read where they are used -- no
// prefetch buffers, nothing of this loop in the registers of the row loops above.  Lanes WITH a slot accumulate y as always.
template <bool ROBUST, bool PK>
__device__ inline void ck_backward_rows_cold(const Dp& d, const CkP& k, const CkRows& R, int row0, int li0, int h, int lane,
                                             const double* P3, const double* lh, const double* lg, bool cold_lane, double* y) {
  for (int j = h - 1; j >= 0; --j) {
    CkStream<2, ROBUST, PK> one;
    one.load(R, row0, li0, j, h, lane, 0);
    const int cp = cold_lane ? k.cpos[(size_t)(row0 + j) * WAVE + lane] : -1;
    const double2 uv = one.uv[0];
    const uint32_t s = (one.w[0] >> (16 * (j & 1))) & 0xffffu;
    if (s != 0xffffu) {
      const double hx = lh[s], hy = lh[s + 1], hz = lh[s + 2];
      const double rw = ROBUST ? ck_huber_w(d, P3, hx, hy, hz, uv) : 1.0;
      const double g[3] = {lg[s], lg[s + 1], lg[s + 2]};
      LplObs o;
      o.set(d, uv, rw);
      double q[3];
      lpl_backward(o, P3, g, q);
      if (cp >= 0) {
        k.q4c[cp] = make_double4(q[0], q[1], q[2], 0.0);
      } else {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          y[4 * m] += hx * q[m];
          y[4 * m + 1] += hy * q[m];
          y[4 * m + 2] += hz * q[m];
          y[4 * m + 3] += q[m];
        }
      }
    }
  }
}

// The lane's camera record, from the rank-ordered image the other E0 kernels stage into LDS (Dp::hot_rec: z (12), then
// P3 row-major (9), 192-byte stride): 168 contiguous bytes = eleven 16-byte loads that touch two or three cache lines.
// (A structure-of-arrays image -- one 8-byte load per entry, neighbouring ranks sharing lines -- was built first: the
// lanes of a tile hold cameras scattered over hundreds of ranks, so each of its 21 + 9 loads per tile touched up to 64
// lines; the texture addresser's work per tile, not HBM, set the kernel's time.)
__device__ inline void ck_load_z(const Dp& d, int rank, double* zz) {
  const double2* r = reinterpret_cast<const double2*>(d.hot_rec + (size_t)rank * HOT_REC_STRIDE);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double2 v = r[j];
    zz[2 * j] = v.x;
    zz[2 * j + 1] = v.y;
  }
}
// ... the Z half from the z image (Dp::zimg: z alone in 96-byte rows by rank -- a row is one or two cache lines that hold
// nothing but z, where the 192-byte records of the other kernels' image straddle three): six 16-byte loads through a buffer
// descriptor built where it is used (four SGPRs that do not live through the row loops).  Plain loads: the hub cameras' rows are
// gathered by every wavefront of every workgroup and live in the L1 -- agent-scope (sc1) loads, which a hand-over of z INSIDE
// the launch would need, cost 3.7 us per term (profiles/r06_experiments.txt D).
__device__ inline __amdgpu_buffer_rsrc_t ck_zimg_rsrc(const Dp& d) {
  return __builtin_amdgcn_make_buffer_rsrc(d.zimg, 0, (unsigned)d.n_cams * 96u, 0x00020000);
}
__device__ inline void ck_load_z_img(const Dp& d, int rank, double* zz) {
  const __amdgpu_buffer_rsrc_t ZR = ck_zimg_rsrc(d);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const ck_u4 a = __builtin_amdgcn_raw_buffer_load_b128(ZR, (unsigned)rank * 96u + 16u * j, 0, 0);
    zz[2 * j] = __hiloint2double((int)a.y, (int)a.x);
    zz[2 * j + 1] = __hiloint2double((int)a.w, (int)a.z);
  }
}
__device__ inline void ck_load_p3(const Dp& d, int rank, double* P3) {
  const double2* r = reinterpret_cast<const double2*>(d.hot_rec + (size_t)rank * HOT_REC_STRIDE) + 6;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double2 v = r[j];
    P3[2 * j] = v.x;
    P3[2 * j + 1] = v.y;
  }
  P3[8] = d.hot_rec[(size_t)rank * HOT_REC_STRIDE + 20];
}
// ... and its translation column (entries 21..23 of the record: build_hot_rec), for ck_huber_w
template <bool ROBUST>
__device__ inline void ck_load_p(const Dp& d, int rank, double* P3) {
  ck_load_p3(d, rank, P3);
  if (ROBUST) {
    const double* r = d.hot_rec + (size_t)rank * HOT_REC_STRIDE;
    P3[9] = r[21];
    const double2 v = reinterpret_cast<const double2*>(r)[11];
    P3[10] = v.x;
    P3[11] = v.y;
  }
}

// end of a tile's backward pass: the chunk sums go to the camera's accumulator in LDS (lanes that share one are summed
// first) or, for a camera without a slot in this workgroup, to the chunk's own partial record
// Partial records leave the kernel through PLAIN stores: the eight L2s gather a record's six 16-byte pieces (and the pieces of
// neighbouring records) into whole lines and the release at the kernel's end writes them back in bulk.  Measured (round 6,
// profiles/r06_experiments.txt B; venice-1778, us per term in the replayed graph): plain 60.4, write-through (sc1, aux 16: every
// 16-byte piece its own fabric write) 64.4, non-temporal (aux 2) 65.0, both 67.0 -- which is also what a hand-over of the
// records INSIDE the launch (per-camera arrival counters, the owner sums) would have to pay before anything else.
#ifndef POVAR_CK_PART_AUX
#define POVAR_CK_PART_AUX 0  // aux bits of the buffer store: 0 = plain, 16 = sc1 (agent scope, write-through), 2 = nt
#endif
__device__ inline __amdgpu_buffer_rsrc_t ck_part_rsrc(double* part_out) {
  return __builtin_amdgcn_make_buffer_rsrc(part_out, 0, 0x7ffffff0, 0x00020000);  // (records: 96 bytes x < 2^24)
}
__device__ inline void ck_store_part(__amdgpu_buffer_rsrc_t pr, unsigned byte_off, double a, double b) {
  ck_u4 g;
  g.x = (unsigned)__double2loint(a); g.y = (unsigned)__double2hiint(a); g.z = (unsigned)__double2loint(b); g.w = (unsigned)__double2hiint(b);
  __builtin_amdgcn_raw_buffer_store_b128(g, pr, byte_off, 0, POVAR_CK_PART_AUX);
}
// end of a tile's backward pass: the chunk sums go to the camera's accumulator in LDS (lanes that share one are summed
// first) or, for a camera without a slot in this workgroup, to the chunk's own partial record
__device__ inline void ck_flush_tile(double (&y)[12], int flags, int lane, int rank, int acc_slot, int seg, double* acc, int n_acc,
                                     double* part_ptr) {
  if (flags & 1) seg_scan_steps<12>(y, lane, seg & 255, 4);  // (inclusive scan: the run's total is in its LAST lane)
  if (rank >= 0) {
    if (acc_slot >= 0) {
      if (lane == ((seg >> 8) & 255)) {
#pragma unroll
        for (int m = 0; m < 12; ++m)  // acc[slot][13]: one address register, twelve immediate offsets; odd stride: 32 bank classes
          __hip_atomic_fetch_add(acc + acc_slot * CK_ACC_STRIDE + m, y[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (part_ptr) {  // (nullptr: the lane has left q of its observations in the cold view: ck_backward_rows_cold)
      const __amdgpu_buffer_rsrc_t part_out = ck_part_rsrc(part_ptr);
      const unsigned o = (unsigned)(~acc_slot) * 96u;
#pragma unroll
      for (int m = 0; m < 6; ++m) ck_store_part(part_out, o + 16u * m, y[2 * m], y[2 * m + 1]);
    }
  }
}

// NW wavefronts per workgroup; SD: rows a tile keeps in flight ahead of the row being worked on.
//
// What bounds the kernel (in-kernel stamps of a diagnostic build, tools/variants/ck_stamps.patch + tools/ck_stamps.py; profiles/r04_*): the row loops issue 38 VALU instructions
// per row and pass (31 fp64) -- and are not what sets the time: 14 % fewer instructions changed nothing --; the rest is the
// time line of ONE wavefront -- round trips that nothing of its own hides (one or two tiles per wavefront and pass):
// tile metadata (which camera is in which lane), the record gather that depends on it (64 lanes, 64 different cache
// lines: ~30 cycles of the texture addresser per load, eleven loads, sixteen wavefronts at once), the first rows.  So:
//   * a wavefront walks its tiles of a batch forward and then back in REVERSE order: the way back starts with the tile
//     whose P3 is still in registers (no second gather for the 12 of 16 wavefronts that have one tile) and whose rows
//     were read last (they come back in reverse too: what the XCD's L2 still holds is read first);
//   * every pass is started before the workgroup barrier in front of it (gather and first rows of the first tile in
//     flight while the wavefront waits), the metadata of a wavefront's next tile is requested before it walks the
//     current one, and the landmark coordinates + first metadata of the NEXT batch are requested before the way back
//     of the current one;
//   * the rounds of the tile walk alternate direction (tile_of): the short tiles of the second round go to the
//     wavefronts with the shortest first tiles -- the longest-first schedule the layout was cut for, without a counter
//     that would hide which tile comes next.
// NG groups of NW / NG wavefronts each (batch b belongs to group b % NG; the layout's batch count is a multiple of NG and the
// LDS holds NG batches of landmark slots).
template <int NW, int SD, bool DB, int NG, bool ROBUST, bool PK>
__global__ __launch_bounds__(NW * 64) void e0_ck(Dp d, CkP k, double* part_out) {
  const int done = d.flags[1];
  extern __shared__ double ck_lds[];
  constexpr int GW = NW / NG;  // wavefronts of a group
  const CkRows R = ck_rows(k);
  const int S = k.slots;
  const int lane0 = threadIdx.x & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = NG > 1 ? wave_all / GW : 0;   // (consecutive wavefronts go to the SIMDs in turn: every SIMD hosts both groups)
  const int wave = NG > 1 ? wave_all % GW : wave_all;  // number inside the group
  int* gbase = reinterpret_cast<int*>(ck_lds);            // [NG] barrier counters of the groups (16 bytes reserved)
  // Slot-major ([S][3], 24-byte stride): the three entries of a landmark are ONE address + immediate offsets (component-
  // major [3][S] cost six address additions per observation and pass in loops that are VALU-bound at 44 instructions
  // per row); slot -> bank pair is still a bijection mod 32 (6 s mod 64): the layout's row placement is unchanged.
  double* lh = ck_lds + 2 + (size_t)grp * 6 * S;  // [S][3] landmark coordinates of the group's batch
  double* lu = lh + 3 * S;                    // [S][3] u = Jl^T Jp x, then g = G u
  double* acc = ck_lds + 2 + (size_t)NG * 6 * S;  // [n_acc][13] per-camera accumulators of the workgroup
  int* gcnt = gbase + grp;
  int ggen = 0;
  const V2& v = d.v2;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_acc = v.wg_cam_off[blockIdx.x + 1] - cam0;
  const int t0 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t1 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)k.tile;
  const cint_p bt = (cint_p)(uintptr_t)k.bt_off;
  if (done) return;  // wave-uniform, before any barrier and any side effect
  auto group_barrier = [&]() {
    if (NG > 1) ck_group_barrier(gcnt, ggen, GW, lane0);
    else ck_barrier();
  };
  // Which tile of a round a wavefront walks: consecutive wavefronts sit on the four SIMDs in turn and a SIMD issues for its oldest
  // wavefront first, so with "wavefront w walks tile w" SIMD 0 hosted the longest tile of every group of four and its wavefronts 4
  // and 8 closed the way forward 4 k cycles behind the median in 225 of 256 workgroups (profiles/r06_e0_ck_phase_stamps.txt).
  // The groups of four are dealt in alternating direction instead: h0 + h7 + h8 + h15 on SIMD 0, h3 + h4 + h11 + h12 on SIMD 3
  // (53.98 -> 53.55 us per term on venice, profiles/r06_snake_ab.txt).
  const int wave_t = (wave & ~3) | (((wave >> 2) & 1) ? 3 - (wave & 3) : (wave & 3));
  auto tile_of = [&](int tb0, int q) { return tb0 + q * GW + ((q & 1) ? GW - 1 - wave_t : wave_t); };
  constexpr int HM = 32 / NW > 0 ? 32 / NW : 1;  // slot tiles per wavefront whose h~ / G are requested a phase ahead
  double hn[HM][3];
  auto request_h = [&](int b, int lane) {
#pragma unroll
    for (int q = 0; q < HM; ++q) {
      hn[q][0] = hn[q][1] = hn[q][2] = 0;
      const int m = wave + q * GW;
      if (b < k.nb && t0 + b + k.nb * m < t1) {
        const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9) * WAVE + lane;
        hn[q][0] = rp[0]; hn[q][1] = rp[WAVE]; hn[q][2] = rp[2 * WAVE];
      }
    }
  };
  // The tile ranges of the workgroup's batches: from LDS after the first batch (sixteen spare ints behind the accumulators),
  // and the next batch's first requests are issued BEHIND the barrier that ends the way forward: every scalar load -- each
  // workgroup's first touch of its bt_off entries is a miss all the way -- counts against the lgkmcnt(0) in front of a
  // barrier (in-kernel stamps: that barrier opened 9.0 k cycles after the last wavefront's last row in the first batch,
  // 4.1 k in the last one, which requests no next batch; now 4.0 k in both)
  int* lbt = reinterpret_cast<int*>(acc + (size_t)k.max_acc * CK_ACC_STRIDE);
  const bool bt_lds = k.nb + 1 <= 16;
  auto bt_of = [&](int i, bool first) {
    return (bt_lds && !first) ? __builtin_amdgcn_readfirstlane(lbt[i]) : bt[blockIdx.x * k.nb + i];
  };
  int rank_next = 0;  // camera ranks of the wavefront's first tile of the next batch (requested with hn)
  auto request_first_meta = [&](int b, int lane) {
    rank_next = 0;
    if (b < k.nb) {
      const int tb0 = bt_of(b, b == grp), tb1 = bt_of(b + 1, b == grp);
      if (tb0 + wave_t < tb1) rank_next = ck_rank(k.lane_meta[(size_t)(tb0 + wave_t) * WAVE + lane].x);
    }
  };
  request_first_meta(grp, lane0);
  request_h(grp, lane0);
  for (int i = threadIdx.x; i < n_acc * CK_ACC_STRIDE; i += NW * 64) acc[i] = 0;
  if (NG > 1 && threadIdx.x < NG) gbase[threadIdx.x] = 0;
  if (bt_lds && (int)threadIdx.x <= k.nb) lbt[threadIdx.x] = k.bt_off[blockIdx.x * k.nb + threadIdx.x];
  if (NG > 1) ck_barrier();  // accumulators and counters are zero before any group goes on (NG = 1: the first batch barrier)
  for (int b = grp; b < k.nb; b += NG) {
    // The lane number is made opaque per batch (and again per pass): every per-lane address of the body (a dozen 64-bit
    // pointers into the row, metadata and record arrays) is otherwise hoisted out of the batch loop as loop-invariant and
    // held in registers through all of it.  And every array is initialised on the path that does not load it: an
    // undefined value makes the compiler carry the PREVIOUS iteration's registers through the whole body instead (42
    // VGPRs held across the way back; the row loop's camera record went to scratch: 8 reloads per row).
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tb0 = bt_of(b, b == grp), tb1 = bt_of(b + 1, b == grp);
    int q_t = 0;  // round of the tile walk
    int t = tile_of(tb0, 0);
    int rank = rank_next;
    double zz[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, P3[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (P3[9..11]: t, HUBER only)
    CkStream<SD, ROBUST, PK> st;
    st.clear();
    int row0 = 0, h = 0, fl = 0, li0 = 0;
    // (Order of the requests: the rows of a tile are requested after everything else of its phase.  The wait counters
    // retire in issue order; with younger loads pending behind the rows, the compiler's merge of the loop-entry and
    // back-edge states at the head of the row loop came out as s_waitcnt vmcnt(0): every row waited for the row
    // requested one step earlier.)
    //
    // DB (double buffering, for instantiations with registers to spare: 12 wavefronts, 168 VGPRs): record and first rows
    // of the tile AFTER the current one are requested before the current one is walked -- a wavefront's second tile
    // then starts without the 8-9 thousand cycles (gather + rows, one dependent round trip under load) that made the
    // four wavefronts with a second tile the tail of every pass.
    int tn = tile_of(tb0, 1);
    int rank_n = 0, rank_nn = 0;
    double zn[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, Pn[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    CkStream<SD, ROBUST, PK> stn;
    stn.clear();
    int row0n = 0, hnx = 0, fln = 0, li0n = 0;
    // ---- the way forward starts: record and first rows of the first tile (the metadata came with the last phase)
    if (t < tb1) {
      row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];
      if (tn < tb1) rank_n = ck_rank(k.lane_meta[(size_t)tn * WAVE + lane].x);
      ck_load_p<ROBUST>(d, rank < 0 ? 0 : rank, P3);
      st.template start<1>(R, row0, li0, h, lane);
    }
    if (t < tb1) ck_load_z_img(d, rank < 0 ? 0 : rank, zz);
    // ---- landmark coordinates of the batch into LDS (requested a phase ago), u = 0
#pragma unroll
    for (int q = 0; q < HM; ++q) {
      const int m = wave + q * GW;
      if (t0 + b + k.nb * m < t1) {
        const int s = m * WAVE + lane;
        lh[3 * s] = hn[q][0];
        lh[3 * s + 1] = hn[q][1];
        lh[3 * s + 2] = hn[q][2];
        lu[3 * s] = 0;
        lu[3 * s + 1] = 0;
        lu[3 * s + 2] = 0;
      }
    }
    for (int m = wave + HM * GW; t0 + b + k.nb * m < t1; m += GW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9) * WAVE + lane;
      const int s = m * WAVE + lane;
      lh[3 * s] = rp[0];
      lh[3 * s + 1] = rp[WAVE];
      lh[3 * s + 2] = rp[2 * WAVE];
      lu[3 * s] = 0;
      lu[3 * s + 1] = 0;
      lu[3 * s + 2] = 0;
    }
    // the tile after the first: its record and first rows (DB), the metadata of the one after that
    auto request_next_fwd = [&]() {
      if (DB && tn < tb1) {
        const int tnn = tile_of(tb0, q_t + 2);
        row0n = tiles[4 * tn]; hnx = tiles[4 * tn + 1]; fln = tiles[4 * tn + 2]; li0n = tiles[4 * tn + 3];
        if (tnn < tb1) rank_nn = ck_rank(k.lane_meta[(size_t)tnn * WAVE + lane].x);
        const int rk = rank_n < 0 ? 0 : rank_n;
        ck_load_z_img(d, rk, zn);
        ck_load_p<ROBUST>(d, rk, Pn);
        stn.template start<1>(R, row0n, li0n, hnx, lane);
      }
    };
    request_next_fwd();
    group_barrier();
    // (Z is the youngest request of the first tile: waited for HERE, once -- left to the row loop's first use the compiler
    // merges the loop's entry and back-edge states into a wait for every outstanding load at the head of each row)
#pragma unroll
    for (int e = 0; e < 12; ++e) asm volatile("" : "+v"(zz[e]));
    // ---- forward
    while (t < tb1) {
      ck_forward_rows<SD, ROBUST, PK>(d, R, st, row0, li0, h, lane, zz, P3, lh, lu, S);
      if (tn >= tb1) break;  // (t, q_t, rank, P3 stay on the last tile: the way back starts there)
      t = tn;
      ++q_t;
      rank = rank_n;
      tn = tile_of(tb0, q_t + 1);
      if (DB) {
#pragma unroll
        for (int e = 0; e < 12; ++e) zz[e] = zn[e];
#pragma unroll
        for (int e = 0; e < (ROBUST ? 12 : 9); ++e) P3[e] = Pn[e];
        st = stn;
        row0 = row0n; h = hnx; fl = fln; li0 = li0n;
        rank_n = rank_nn;
        request_next_fwd();
      } else {
        row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];
        if (tn < tb1) rank_n = ck_rank(k.lane_meta[(size_t)tn * WAVE + lane].x);
        const int rk = rank < 0 ? 0 : rank;
        ck_load_z_img(d, rk, zz);
        ck_load_p<ROBUST>(d, rk, P3);
        st.template start<1>(R, row0, li0, h, lane);
      }
    }
    // ---- the way back starts before the barriers in front of it: accumulator metadata and last rows of the tile the
    // wavefront has just left (its P3 is in registers), G of its landmark slots, and the next batch's first requests
    asm volatile("" : "+v"(lane));  // (the two passes share no per-lane address register)
    int acc_slot = 0, seg = 0;
    int tp = q_t > 0 ? tile_of(tb0, q_t - 1) : tb1;
    int rank_p = 0, acc_p = 0, seg_p = 0, rank_pp = 0, acc_pp = 0, seg_pp = 0;
    double G[HM][6];
#pragma unroll
    for (int q = 0; q < HM; ++q) {
#pragma unroll
      for (int e = 0; e < 6; ++e) G[q][e] = 0;
      const int m = wave + q * GW;
      if (t0 + b + k.nb * m < t1) {
        const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9 + 3) * WAVE + lane;
#pragma unroll
        for (int e = 0; e < 6; ++e) G[q][e] = rp[e * WAVE];
      }
    }
    if (t < tb1) {
      const int2 me = k.lane_meta[(size_t)t * WAVE + lane];
      seg = ck_seg(me.x);
      acc_slot = me.y;  // (fl: the tile's header word came with row0 / h / li0 -- a scalar load HERE sits in front of the barrier's
      if (tp < tb1) {   //  lgkmcnt(0): 2.6 k cycles per batch in the stamps of round 6, profiles/r06_e0_ck_phase_stamps.txt)
        const int2 mp = k.lane_meta[(size_t)tp * WAVE + lane];
        rank_p = ck_rank(mp.x);
        seg_p = ck_seg(mp.x);
        acc_p = mp.y;
      }
      st.template start<-1>(R, row0, li0, h, lane);
    }
    group_barrier();
    request_first_meta(b + NG, lane);  // the next batch is started from here: its coordinates and first metadata are in
    request_h(b + NG, lane);           // flight during the way back (behind the barrier: see lbt)
    // ---- g = G u per landmark slot (over u)
#pragma unroll
    for (int q = 0; q < HM; ++q) {
      const int m = wave + q * GW;
      if (t0 + b + k.nb * m < t1) {
        const int s = m * WAVE + lane;
        const double u0 = lu[3 * s], u1 = lu[3 * s + 1], u2 = lu[3 * s + 2];
        lu[3 * s] = G[q][0] * u0 + G[q][1] * u1 + G[q][2] * u2;
        lu[3 * s + 1] = G[q][1] * u0 + G[q][3] * u1 + G[q][4] * u2;
        lu[3 * s + 2] = G[q][2] * u0 + G[q][4] * u1 + G[q][5] * u2;
      }
    }
    for (int m = wave + HM * GW; t0 + b + k.nb * m < t1; m += GW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9 + 3) * WAVE + lane;
      const double g0 = rp[0], g1 = rp[WAVE], g2 = rp[2 * WAVE], g3 = rp[3 * WAVE], g4 = rp[4 * WAVE], g5 = rp[5 * WAVE];
      const int s = m * WAVE + lane;
      const double u0 = lu[3 * s], u1 = lu[3 * s + 1], u2 = lu[3 * s + 2];
      lu[3 * s] = g0 * u0 + g1 * u1 + g2 * u2;
      lu[3 * s + 1] = g1 * u0 + g3 * u1 + g4 * u2;
      lu[3 * s + 2] = g2 * u0 + g4 * u1 + g5 * u2;
    }
    // the tile before the current one on the way back: its P3 and last rows (DB), the metadata of the one before that
    auto request_next_bwd = [&]() {
      if (DB && tp < tb1) {
        const int tpp = q_t > 1 ? tile_of(tb0, q_t - 2) : tb1;
        row0n = tiles[4 * tp]; hnx = tiles[4 * tp + 1]; fln = tiles[4 * tp + 2]; li0n = tiles[4 * tp + 3];
        if (tpp < tb1) {
          const int2 mp = k.lane_meta[(size_t)tpp * WAVE + lane];
          rank_pp = ck_rank(mp.x);
          seg_pp = ck_seg(mp.x);
          acc_pp = mp.y;
        }
        ck_load_p<ROBUST>(d, rank_p < 0 ? 0 : rank_p, Pn);
        stn.template start<-1>(R, row0n, li0n, hnx, lane);
      }
    };
    request_next_bwd();
    group_barrier();
    // ---- backward: the wavefront's tiles in reverse
    while (t < tb1) {
      double y[12];
#pragma unroll
      for (int m = 0; m < 12; ++m) y[m] = 0;
      // (wave-uniform; 2 = CK_FLAG_COLD.  Not in the instantiations that recompute a robust weight: they have no registers for the
      //  cold loop beside their row loops -- ten spilled per lane at batch level, 30 MB of scratch traffic per launch -- and gain
      //  nothing from the cold view; povar_create.hip keeps the records for them)
      const bool cold_q = !ROBUST && k.cpos != nullptr && (fl & 2) != 0;
      if (cold_q) ck_backward_rows_cold<ROBUST, PK>(d, k, R, row0, li0, h, lane, P3, lh, lu, rank >= 0 && acc_slot < 0, y);
      else ck_backward_rows<SD, ROBUST, PK>(d, R, st, row0, li0, h, lane, P3, lh, lu, S, y);
      ck_flush_tile(y, fl, lane, rank, acc_slot, seg, acc, n_acc, cold_q ? nullptr : part_out);
      if (tp >= tb1) break;
      t = tp;
      --q_t;
      rank = rank_p; acc_slot = acc_p; seg = seg_p;
      tp = q_t > 0 ? tile_of(tb0, q_t - 1) : tb1;
      if (DB) {
#pragma unroll
        for (int e = 0; e < (ROBUST ? 12 : 9); ++e) P3[e] = Pn[e];
        st = stn;
        row0 = row0n; h = hnx; fl = fln; li0 = li0n;
        rank_p = rank_pp; acc_p = acc_pp; seg_p = seg_pp;
        request_next_bwd();
      } else {
        row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];
        if (tp < tb1) {
          const int2 mp = k.lane_meta[(size_t)tp * WAVE + lane];
          rank_p = ck_rank(mp.x);
          seg_p = ck_seg(mp.x);
          acc_p = mp.y;
        }
        ck_load_p<ROBUST>(d, rank < 0 ? 0 : rank, P3);
        st.template start<-1>(R, row0, li0, h, lane);
      }
    }
    group_barrier();  // the next batch overwrites h~ and u; after the last one: the accumulators are complete
  }
  if (NG > 1) ck_barrier();  // every group is done: the accumulators are complete
  // ---- accumulators -> this workgroup's partial records (camera-major in part_out)
  const __amdgpu_buffer_rsrc_t PR = ck_part_rsrc(part_out);
  for (int i = threadIdx.x; i < n_acc * 6; i += NW * 64) {
    const int r = i / 6, m = 2 * (i % 6);
    const int rec = k.slot_rec[cam0 + r];
    ck_store_part(PR, (unsigned)rec * 96u + 16u * (unsigned)(i % 6), acc[r * CK_ACC_STRIDE + m], acc[r * CK_ACC_STRIDE + m + 1]);
  }
  if (d.p2p_epoch && blockIdx.x == 0 && threadIdx.x == 0) *d.p2p_epoch += 1;  // one tick per term (as e0_lpl)
}

// robust weights in chunk order: w_ck[i] = w_lpl[src[i]] (once per linearisation; V2::w is written by lpl_pass<0>)
POVAR_KERNEL __launch_bounds__(256) void ck_gather_w(const int* src, const double* w_lpl, double* w_ck, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = src[i];
  w_ck[i] = s >= 0 ? w_lpl[s] : 0.0;
}

}  // namespace povar
