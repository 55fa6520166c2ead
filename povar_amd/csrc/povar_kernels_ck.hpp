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
// Every row (18 bytes per observation) is read on both passes.  (Keeping the rows of a batch in registers between the
// passes was built and measured: at the four wavefronts per SIMD the kernel needs for latency there are 128 VGPRs per
// lane, the forward pass alone needs ~120 of them, and with 8 or 12 wavefronts per workgroup it ran twice as long.)
#pragma once

#include "povar_kernels.hpp"

namespace povar {

constexpr int CK_ROWS = 16;  // = CK_HMAX of ck_layout.hpp: rows per chunk tile at most

struct CkP {
  const double2* uv;     // [rows][64]
  const uint32_t* li;    // [li_rows][64] two 16-bit landmark slots per word
  const double* w;       // [rows][64] robust weights (only with a robust norm)
  const int4* tile;      // first row, height, flags, first li row
  const int* lane_cam;   // [tiles][64] rank of the lane's camera (-1: empty lane)
  const int* lane_acc;   // [tiles][64] >= 0 accumulator slot, < 0: ~(partial record of a cold chunk)
  const int* lane_seg;   // [tiles][64] first | last << 8 lane sharing the accumulator
  const int* bt_off;     // [grid * nb + 1]
  const int* slot_rec;   // partial record of each workgroup slot
  const double* img;     // [21][pad] structure-of-arrays record image by rank: z (12), then P3 row-major (9)
  int nb, slots, pad;
};

constexpr int CK_ACC_STRIDE = 13;  // doubles per accumulator slot in LDS (12 used)
__host__ __device__ inline size_t ck_lds_bytes_dev(int slots, int n_acc) { return (size_t)slots * 48 + (size_t)n_acc * CK_ACC_STRIDE * 8 + 64; }

// One observation forward: u_l += P3^T (w C (Z h~_l)); backward: y_c += h~_l (x) (w C (P3 g_l))
__device__ inline void ck_obs_forward(const Dp& d, double2 uv, double w, const double* zz, const double* P3, double hx, double hy,
                                      double hz, double* lu, int S, uint32_t s) {
  LplObs o;
  o.set(d, uv, w);
  double red[3] = {0, 0, 0};
  lpl_forward(o, zz, P3, hx, hy, hz, red);
  __hip_atomic_fetch_add(lu + s, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(lu + S + s, red[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(lu + 2 * S + s, red[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// Streamed tile (rows read where they are used): D rows are in flight ahead of the row being worked on (the kernel is
// bound by the bytes it keeps in flight: 16 wavefronts x 2 rows x 1.1 KB per CU sustain 4 TB/s, not 6).  A rolled loop
// over the rows with a shift register of D row buffers (D - 1 register moves per row: the unrolled-by-D form with
// statically indexed buffers made the compiler hoist and spill 160 VGPRs).
template <int D, bool ROBUST>
struct CkStream {
  double2 uv[D];
  uint32_t w[D];
  double rw[D];
  __device__ inline void load(const CkP& k, int row0, int li0, int j, int h, int lane, int i) {
    if (j < h) {
      uv[i] = k.uv[((size_t)row0 + j) * WAVE + lane];
      w[i] = k.li[((size_t)li0 + (j >> 1)) * WAVE + lane];
      if (ROBUST) rw[i] = k.w[((size_t)row0 + j) * WAVE + lane];
    }
  }
  __device__ inline void clear() {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      uv[i] = make_double2(0, 0);
      w[i] = 0xffffffffu;
      rw[i] = 1.0;
    }
  }
  __device__ inline void start(const CkP& k, int row0, int li0, int h, int lane) {
    clear();
#pragma unroll
    for (int i = 0; i < D; ++i) load(k, row0, li0, i, h, lane, i);
  }
  // row j leaves the register, row j + D is requested
  __device__ inline void next(const CkP& k, int row0, int li0, int j, int h, int lane, double2& uv_j, uint32_t& s_j, double& rw_j) {
    uv_j = uv[0];
    s_j = (w[0] >> (16 * (j & 1))) & 0xffffu;
    rw_j = rw[0];
#pragma unroll
    for (int i = 0; i + 1 < D; ++i) {
      uv[i] = uv[i + 1];
      w[i] = w[i + 1];
      if (ROBUST) rw[i] = rw[i + 1];
    }
    load(k, row0, li0, j + D, h, lane, D - 1);
  }
};
// the rows of one tile, forward / backward; st has been started on the tile (its first D rows are in flight)
template <int D, bool ROBUST>
__device__ inline void ck_forward_rows(const Dp& d, const CkP& k, CkStream<D, ROBUST>& st, int row0, int li0, int h, int lane,
                                       const double* zz, const double* P3, const double* lh, double* lu, int S) {
#pragma nounroll
  for (int j = 0; j < h; ++j) {
    double2 uv;
    uint32_t s;
    double rw;
    st.next(k, row0, li0, j, h, lane, uv, s, rw);
    if (s != 0xffffu) {
      const double hx = lh[s], hy = lh[S + s], hz = lh[2 * S + s];
      ck_obs_forward(d, uv, ROBUST ? rw : 1.0, zz, P3, hx, hy, hz, lu, S, s);
    }
  }
}
template <int D, bool ROBUST>
__device__ inline void ck_backward_rows(const Dp& d, const CkP& k, CkStream<D, ROBUST>& st, int row0, int li0, int h, int lane,
                                        const double* P3, const double* lh, const double* lg, int S, double* y) {
#pragma nounroll
  for (int j = 0; j < h; ++j) {
    double2 uv;
    uint32_t s;
    double rw;
    st.next(k, row0, li0, j, h, lane, uv, s, rw);
    if (s != 0xffffu) {
      const double hx = lh[s], hy = lh[S + s], hz = lh[2 * S + s];
      const double g[3] = {lg[s], lg[S + s], lg[2 * S + s]};
      ck_obs_backward(d, uv, ROBUST ? rw : 1.0, P3, hx, hy, hz, g, y);
    }
  }
}

__device__ inline void ck_load_z(const CkP& k, int rank, double* zz) {
#pragma unroll
  for (int j = 0; j < 12; ++j) zz[j] = k.img[(size_t)j * k.pad + rank];
}
__device__ inline void ck_load_p3(const CkP& k, int rank, double* P3) {
#pragma unroll
  for (int j = 0; j < 9; ++j) P3[j] = k.img[(size_t)(12 + j) * k.pad + rank];
}

// end of a tile's backward pass: the chunk sums go to the camera's accumulator in LDS (lanes that share one are summed
// first) or, for a camera without a slot in this workgroup, to the chunk's own partial record
__device__ inline void ck_flush_tile(double (&y)[12], int flags, int lane, int rank, int acc_slot, int seg, double* acc, int n_acc,
                                     double* part_out) {
  if (flags & 1) seg_reduce_steps<12>(y, lane, seg & 255, (seg >> 8) & 255, 4);
  if (rank >= 0) {
    if (acc_slot >= 0) {
      if (lane == (seg & 255)) {
#pragma unroll
        for (int m = 0; m < 12; ++m)  // acc[slot][13]: one address register, twelve immediate offsets; odd stride: 32 bank classes
          __hip_atomic_fetch_add(acc + acc_slot * CK_ACC_STRIDE + m, y[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
      double2* o = reinterpret_cast<double2*>(part_out + (size_t)(~acc_slot) * 12);
#pragma unroll
      for (int m = 0; m < 6; ++m) o[m] = make_double2(y[2 * m], y[2 * m + 1]);
    }
  }
}

// NW wavefronts per workgroup; SD: rows a tile keeps in flight ahead of the row being worked on.
// A wavefront's time line is a chain of round trips -- tile metadata (which camera is in which lane), then the record
// gather that depends on it, then the rows -- and with one or two tiles per wavefront and pass nothing else of its own
// hides them.  So every pass is started BEFORE the workgroup barrier in front of it: the metadata of the wavefront's
// first tile is requested first, the gather and the first SD rows follow as soon as it is there and are in flight while
// the wavefront waits at the barrier; the metadata of a wavefront's next tile is requested before it walks the current one.
template <int NW, int SD, bool ROBUST>
__global__ __launch_bounds__(NW * 64) void e0_ck(Dp d, CkP k, double* part_out) {
  const int done = d.flags[1];
  extern __shared__ double ck_lds[];
  const V2& v = d.v2;
  const int S = k.slots;
  double* lh = ck_lds;            // [3][S] landmark coordinates of the batch
  double* lu = ck_lds + 3 * S;    // [3][S] u = Jl^T Jp x, then g = G u
  double* acc = ck_lds + 6 * S;   // [n_acc][13] per-camera accumulators of the workgroup
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_acc = v.wg_cam_off[blockIdx.x + 1] - cam0;
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int t0 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x]);
  const int t1 = __builtin_amdgcn_readfirstlane(v.wg_tile_off[blockIdx.x + 1]);
  for (int i = threadIdx.x; i < n_acc * CK_ACC_STRIDE; i += NW * 64) acc[i] = 0;
  typedef const int __attribute__((address_space(4))) * cint_p;
  const cint_p tiles = (cint_p)(uintptr_t)k.tile;
  const cint_p bt = (cint_p)(uintptr_t)k.bt_off;
  if (done) return;  // wave-uniform, before any barrier and any side effect
  for (int b = 0; b < k.nb; ++b) {
    // The lane number is made opaque per batch: every per-lane address of the body (a dozen 64-bit pointers into the row,
    // metadata and record arrays) is otherwise hoisted out of the batch loop as loop-invariant and held in registers
    // through all of it -- 30-40 VGPRs the row loops then lack (their camera record went to scratch: 8 reloads per row).
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tb0 = bt[blockIdx.x * k.nb + b], tb1 = bt[blockIdx.x * k.nb + b + 1];
    int t = tb0 + wave;
    int rank = 0;
    if (t < tb1) rank = k.lane_cam[(size_t)t * WAVE + lane];
    // ---- landmark coordinates of the batch into LDS, u = 0
    for (int m = wave; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9) * WAVE + lane;
      const int s = m * WAVE + lane;
      lh[s] = rp[0];
      lh[S + s] = rp[WAVE];
      lh[2 * S + s] = rp[2 * WAVE];
      lu[s] = 0;
      lu[S + s] = 0;
      lu[2 * S + s] = 0;
    }
    // ---- forward
    {
      // (every array below is initialised on the path that does not load it: an undefined value makes the compiler carry
      // the PREVIOUS batch iteration's registers through the whole loop body instead -- 42 VGPRs held across the way back)
      double zz[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, P3[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      CkStream<SD, ROBUST> st;
      st.clear();
      int row0 = 0, h = 0, li0 = 0;
      if (t < tb1) {
        row0 = tiles[4 * t]; h = tiles[4 * t + 1]; li0 = tiles[4 * t + 3];
        const int rk = rank < 0 ? 0 : rank;
        ck_load_z(k, rk, zz);
        ck_load_p3(k, rk, P3);
        st.start(k, row0, li0, h, lane);
      }
      __syncthreads();
      while (t < tb1) {
        const int tn = t + NW;
        int rank_n = 0;
        if (tn < tb1) rank_n = k.lane_cam[(size_t)tn * WAVE + lane];
        ck_forward_rows<SD, ROBUST>(d, k, st, row0, li0, h, lane, zz, P3, lh, lu, S);
        t = tn;
        if (t < tb1) {
          row0 = tiles[4 * t]; h = tiles[4 * t + 1]; li0 = tiles[4 * t + 3];
          const int rk = rank_n < 0 ? 0 : rank_n;
          ck_load_z(k, rk, zz);
          ck_load_p3(k, rk, P3);
          st.start(k, row0, li0, h, lane);
        }
      }
    }
    // ---- the way back is started before the barriers in front of it: metadata, P3 and first rows of the wavefront's
    // first tile, and G of its landmark slots (NW x GM slot tiles without a loop, the rest in the loop below)
    asm volatile("" : "+v"(lane));  // (again: the two passes share no per-lane address register)
    t = tb0 + wave;
    int acc_slot = 0, seg = 0;
    if (t < tb1) {
      rank = k.lane_cam[(size_t)t * WAVE + lane];
      acc_slot = k.lane_acc[(size_t)t * WAVE + lane];
      seg = k.lane_seg[(size_t)t * WAVE + lane];
    }
#ifndef CK_GM
#define CK_GM (32 / NW > 0 ? 32 / NW : 1)
#endif
    constexpr int GM = CK_GM > 0 ? CK_GM : 1;
    double G[GM][6];
#pragma unroll
    for (int q = 0; q < GM; ++q) {
#pragma unroll
      for (int e = 0; e < 6; ++e) G[q][e] = 0;
      const int m = wave + q * NW;
      if (CK_GM > 0 && t0 + b + k.nb * m < t1) {
        const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9 + 3) * WAVE + lane;
#pragma unroll
        for (int e = 0; e < 6; ++e) G[q][e] = rp[e * WAVE];
      }
    }
    double P3[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    CkStream<SD, ROBUST> st;
    st.clear();
    int row0 = 0, h = 0, fl = 0, li0 = 0;
    if (t < tb1) {
      row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];
      ck_load_p3(k, rank < 0 ? 0 : rank, P3);
      st.start(k, row0, li0, h, lane);
    }
    __syncthreads();
    // ---- g = G u per landmark slot (over u)
#pragma unroll
    for (int q = 0; q < GM; ++q) {
      const int m = wave + q * NW;
      if (CK_GM > 0 && t0 + b + k.nb * m < t1) {
        const int s = m * WAVE + lane;
        const double u0 = lu[s], u1 = lu[S + s], u2 = lu[2 * S + s];
        lu[s] = G[q][0] * u0 + G[q][1] * u1 + G[q][2] * u2;
        lu[S + s] = G[q][1] * u0 + G[q][3] * u1 + G[q][4] * u2;
        lu[2 * S + s] = G[q][2] * u0 + G[q][4] * u1 + G[q][5] * u2;
      }
    }
    for (int m = wave + CK_GM * NW; t0 + b + k.nb * m < t1; m += NW) {
      const double* rp = v.lmrec + ((size_t)(t0 + b + k.nb * m) * 9 + 3) * WAVE + lane;
      const double g0 = rp[0], g1 = rp[WAVE], g2 = rp[2 * WAVE], g3 = rp[3 * WAVE], g4 = rp[4 * WAVE], g5 = rp[5 * WAVE];
      const int s = m * WAVE + lane;
      const double u0 = lu[s], u1 = lu[S + s], u2 = lu[2 * S + s];
      lu[s] = g0 * u0 + g1 * u1 + g2 * u2;
      lu[S + s] = g1 * u0 + g3 * u1 + g4 * u2;
      lu[2 * S + s] = g2 * u0 + g4 * u1 + g5 * u2;
    }
    __syncthreads();
    // ---- backward
    while (t < tb1) {
      const int tn = t + NW;
      int rank_n = 0, acc_n = 0, seg_n = 0;
      if (tn < tb1) {
        rank_n = k.lane_cam[(size_t)tn * WAVE + lane];
        acc_n = k.lane_acc[(size_t)tn * WAVE + lane];
        seg_n = k.lane_seg[(size_t)tn * WAVE + lane];
      }
      double y[12];
#pragma unroll
      for (int m = 0; m < 12; ++m) y[m] = 0;
      ck_backward_rows<SD, ROBUST>(d, k, st, row0, li0, h, lane, P3, lh, lu, S, y);
      ck_flush_tile(y, fl, lane, rank, acc_slot, seg, acc, n_acc, part_out);
      t = tn;
      if (t < tb1) {
        rank = rank_n; acc_slot = acc_n; seg = seg_n;
        row0 = tiles[4 * t]; h = tiles[4 * t + 1]; fl = tiles[4 * t + 2]; li0 = tiles[4 * t + 3];
        ck_load_p3(k, rank < 0 ? 0 : rank, P3);
        st.start(k, row0, li0, h, lane);
      }
    }
    __syncthreads();  // the next batch overwrites h~ and u; after the last one: the accumulators are complete
  }
  // ---- accumulators -> this workgroup's partial records (camera-major in part_out)
  for (int i = threadIdx.x; i < n_acc * 6; i += NW * 64) {
    const int r = i / 6, m = 2 * (i % 6);
    const int rec = k.slot_rec[cam0 + r];
    reinterpret_cast<double2*>(part_out + (size_t)rec * 12)[i % 6] = make_double2(acc[r * CK_ACC_STRIDE + m], acc[r * CK_ACC_STRIDE + m + 1]);
  }
  if (d.p2p_epoch && blockIdx.x == 0 && threadIdx.x == 0) *d.p2p_epoch += 1;  // one tick per term (as e0_lpl)
}

// robust weights in chunk order: w_ck[i] = w_lpl[src[i]] (once per linearisation; V2::w is written by lpl_pass<0>)
__global__ __launch_bounds__(256) void ck_gather_w(const int* src, const double* w_lpl, double* w_ck, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = src[i];
  w_ck[i] = s >= 0 ? w_lpl[s] : 0.0;
}

// structure-of-arrays mirror of the record image: the static part (P3) per linearisation
__global__ __launch_bounds__(256) void ck_build_img_p3(Dp d, double* img, int pad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.n_cams * 9) return;
  const int r = i / 9, e = i % 9;
  const double* P = reinterpret_cast<const double*>(d.cams_lin4) + 12 * (size_t)d.hot_cams[r];
  img[(size_t)(12 + e) * pad + r] = P[(e / 3) * 4 + (e % 3)];
}

}  // namespace povar
