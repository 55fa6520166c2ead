// povar_kernels.hpp -- gfx950 device code for the power-series Schur-complement path (step 1).
//
// Mapping of the reference's TBB regions onto kernels.  Work decomposition:
//
//  * LM ("landmark-major") kernels: one lane per observation.  Observations live in 64-wide
//    wave bins that hold WHOLE landmarks (host packs them, povar_create.hip: build_layout), so the
//    per-landmark 3x3 reductions (Jl^T Jl, Jl^T r, Jl^T t) are wavefront segmented scans over
//    __shfl_up -- no LDS, no barriers, no atomics.  Landmarks with more than 64 observations
//    take the lm_long driver (one workgroup per landmark, LDS block reduction).
//  * CM ("camera-major") kernels: every per-camera sum the reference guards with
//    std::mutex (linearization_power_varproj.hpp:393-397, landmark_block.hpp:531-537) is a
//    gather over a camera->observation inverse index, cut into work items of <= 512
//    observations of ONE camera, one wavefront per item, fixed summation order: deterministic,
//    atomics-free.
//  * camera kernels: 12x12 block work per camera (B^-1 build, B^-1 x, AXPY, norms).
//
// "Implicit" operator form.  The stored tile of one observation (landmark_block.hpp:167-169
// after scale_Jl_cols/scale_Jp_cols, :284-295, :324-334) is a closed-form function of
// (P_c, x_l, u, v, sqrt(w), sigma_c, s_l) (bal_bundle_adjustment_helper.cpp:244-313):
//     Jp = sw * [ sb*(h,0,-u h) ; sb*(0,h,-v h) ; sa*(h,0,0) ; sa*(0,h,0) ] * diag(sigma_c)
//     Jl = sw * [ sb*(P0-uP2) ; sb*(P1-vP2) ; sa*P0 ; sa*P1 ][:, :3] * diag(s_l)
// with h=[x_l;1], sa=sqrt(alpha), sb=sqrt(1-alpha).  Hence
//     Jp x      = sw*( sb*(d0-u d2), sb*(d1-v d2), sa*d0, sa*d1 ),   d_k = h . (sigma*x)[4k:4k+4]
//     Jp^T s    = sigma * ( h q0 ; h q1 ; h q2 ),  q0 = sw(sb s0+sa s2), q1 = sw(sb s1+sa s3),
//                                                  q2 = -sw sb (u s0 + v s1)
// so the E0 product needs 28 B of per-observation input instead of the 480 B tile, and the
// transpose-scatter needs three scalars per observation.  The stored-tile variant
// (POVAR_E0_TILES) keeps the reference's tiles in HBM and streams them once per term.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Kernels that are not templates have internal linkage: the headers are included by every translation unit of the library
// (povar_create.hip, povar_lm.hip, povar_series.hip, ...), each of which compiles the kernels it launches.
#define POVAR_KERNEL static __attribute__((unused)) __global__

namespace povar {

constexpr int WAVE = 64;
constexpr int LM_BLOCK = 256;   // 4 wave bins per workgroup
constexpr int CM_ITEM_MAX = 512; // observations of one camera per CM work item
constexpr int CM_COLD_ITEM_MAX = 128;  // same for the small "cold" view of the LDSACC mode (more, shorter waves)
constexpr int TILE_PAIRS = 32;  // double2 pairs per observation in the blocked tile layout

// meta[slot]: bits 0-7 seg_first lane, 8-15 seg_last lane, 16 = real observation, 17 = slot of
// a long (>64 obs) landmark, bits 18-31 = 1 + rank of the camera in the LDS camera cache (0: not cached)
constexpr int META_REAL = 1 << 16;
constexpr int META_LONG = 1 << 17;
constexpr int HOT_REC_STRIDE = 24;  // doubles per camera in Dp::hot_rec
constexpr int META_HOT_SHIFT = 18;     // 10 bits
constexpr int META_HOT_MASK = 1023;
constexpr int META_STEPS_SHIFT = 28;   // 3 bits: ceil(log2(longest landmark of the bin)), same in every lane of a bin
#ifndef POVAR_HOT_ACC_MAX
#define POVAR_HOT_ACC_MAX 520
#endif
constexpr int HOT_ACC_MAX = POVAR_HOT_ACC_MAX;    // camera slots of a workgroup's LDS: step 2 520 * (208 + 96) B + 48 hub replicas * 96 B
                                    // + 16 B = 163 600 of the 163 840 bytes; step 1 (176-byte records) 154 768
constexpr int HOT_MAX = 912;        // cameras cached per workgroup: 912 * 176 B = 156.75 KiB of the 160 KiB LDS
constexpr int HOT_REC = 11;         // double2 per cached camera: z (6) + P[:, :3] (4.5) + pad
#ifndef POVAR_E0C_BLOCK
#define POVAR_E0C_BLOCK 1024
#endif
constexpr int E0C_BLOCK = POVAR_E0C_BLOCK;  // one workgroup per CU

constexpr int E0_SLOT_BYTES = 32;   // e0_lm_cached<true> per-slot stream: uv 16 + meta 4 + cam 4 + lm 4 + cold_pos 4
constexpr int E0_LMREC_BYTES = 96;  // packed landmark record read by the per-term kernel

// one camera-major index structure: observations sorted by camera, cut into work items
struct CmView {
  const int* slot;         // [n] slot of the p-th observation in camera order
  const double* h;         // [3][n] landmark x,y,z in the same order (per linearisation)
  int64_t n;
  const int* item_off;     // [n_items + 1]
  const int* cam_item_off; // [n_cams + 1]
  double* part;            // [n_items][12]
  int n_items;
  const int2* cam_range;   // [n_cams] (first, end) position of each camera's run (cold view only, else nullptr)
  const int* src;          // [n] where q of the p-th observation is in q4c (the lane-per-landmark kernels store row-major:
                           // lpl_cold_q), nullptr: at p itself (the lane-per-observation kernels store camera-major)
};

// Lane-per-landmark layout of the per-term E0 kernel (e0_lpl).  Landmarks are sorted by (observation count,
// number of cold observations) and cut into tiles of 64 landmarks: lane = landmark, the wavefront loops over
// the observation rows of its tile.  Row r of the arrays holds observation j of every landmark of the tile
// ([row][lane]: one contiguous 1 KiB / 256 B line group per wave instruction).  Inside a landmark the
// observations of LDS-accumulated cameras come first, so the leading rows of a tile are branch-free.
// A landmark with more than K0 (8) observations is dealt over several adjacent lanes of one tile (at most K0 rows
// per tile, whatever the track length: the tiles are the unit of load balance); its partial sums are combined by
// one segmented wavefront scan per tile.
struct V2 {
  const double2* uv;   // [n_rows][64] observation (u, v)
  const int* cw;       // [n_rows][64] >= 0: LDS slot of the camera in this workgroup; -1: no observation;
                       // <= -2: cold observation of the camera with popularity rank -2 - cw (record gathered from L2)
  const int* cpos;     // [n_rows][64] position in the cold camera-major view (-1: not cold)
  double* w;           // [n_rows][64] robust weight (only with a robust norm)
  const int4* tile;    // [n_tiles] (first row, rows, leading all-hot rows, bit 0: some landmark spans several lanes)
  const int* seg;      // [n_tiles][64] first | last << 8 lane of the landmark a lane belongs to (read for flagged tiles)
  double* lmrec;       // [n_tiles][9][64]: x, y, z, then G = diag(s) Hll^-1 diag(s) (00,01,02,11,12,22)
  const int* lm_of;    // [n_tiles][64] landmark of each lane (-1: unused lane)
  // lane-ordered mirrors of the per-landmark arrays (Dp::lms4, lms_lin4, jl_scale4 stay the masters, in landmark order):
  // one coalesced, prefetchable 32-byte load per lane and tile instead of an index load + a 32-byte gather that drags
  // 64..128-byte lines through the memory system.  lmx follows lms4 (povar_lm.hip: ensure_lmx, rebuilt by lm_to_lanes
  // when a landmark writer outside these kernels has run), lml / lsc belong to the linearisation.
  double4* lmx;        // [n_tiles][64] current landmark of each lane
  double4* lml;        // [n_tiles][64] landmark at the linearisation point
  double4* lsc;        // [n_tiles][64] Jl column scale
  const int* lm_pos;   // [n_lms] tile * 64 + first lane of each landmark | (lanes - 1) << 26 (-1: no observation)
  const int* of_slot;  // [n_slots] row * 64 + lane of each wave-bin slot (-1: padding)
  // per E0 workgroup (lpl_layout.hpp): its tiles, the cameras it keeps in LDS, where their accumulators are flushed
  const int* wg_tile_off;  // [grid + 1] tiles of workgroup w, longest first
  const int* wg_cam_off;   // [grid + 1] camera slots of workgroup w
  const int* wg_cams;      // popularity rank (= index in the record image Dp::hot_rec) of the camera in each slot
  const int* wg_slot_rec;  // partial record (12 doubles in hot_out, camera-major) each slot is flushed to
  int n_tiles;
  int hubs;                // leading slots with four accumulator replicas each (lpl_acc_slot)
};

struct Dp {
  int n_cams, n_lms, n_bins, n_items, n_long, n_reg_blocks;
  V2 v2;
  // static landmark-major slot arrays
  const double2* uv;
  const int* cam;
  const int* lm;
  const int* meta;
  int q_rows;           // lane-per-landmark kernels: cold observations leave q row-major (lpl_cold_q, gathered through CmView::src)
  int lin_aux_only;     // OpLinearize: write only the per-slot sqrt(w) / weighted residual (legacy arrays, filled lazily)
  int prep_aux_only;    // OpPrepare[H]: leave the lane-per-landmark records alone (called to fill hll_inv / lmrec lazily)
  int prep_lpl_only;    // prepare_lpl[_h]: write only the lane-per-landmark records, not hll_inv / lmrec (filled lazily)
  const int* hot_cams;  // cameras cached in LDS by e0_lm_cached, most observed first
  double* hot_rec;      // [HOT_MAX][24] contiguous LDS image of the hot cameras' records: z_c (12, rewritten by
                        // every B^-1 kernel) then the static camera part (step 1: P[:, :3] (9), step 2: P (12))
  double* zimg;         // [n_cams][12] z_c alone, by rank (step 1): what e0_ck gathers Z from -- lines of its own, so that the
                        // per-camera step at e0_ck's head can hand z over inside the launch (sc1 stores, sc1 loads only)
  int n_hot;
  const int* long_lm;
  const int* long_first;
  const int* long_cnt;
  const int* lm_slot0;  // [n_lms] first wave-bin slot of each landmark (its slots are consecutive)
  const int* lm_cnt;    // [n_lms] number of observations
  // static camera-major arrays
  const int* cm_slot;
  const int* cm_lm;
  const double2* cm_uv;
  double* cm_h;        // [3][n_obs] landmark x,y,z in camera-major order (per linearisation)
  int64_t n_obs;
  const int* item_off;
  const int* item_cam;
  const int* cam_item_off;
  CmView cmv;            // what cm_scatter and the per-camera item sums walk
  // LDS-accumulated partial sums of the hottest cameras (POVAR_E0_IMPLICIT_LDSACC), else nullptr
  const double* hot_part;  // [n_hot_acc][n_hot_wg][12], or partial records addressed through part_range
  const int2* part_range;  // [n_cams] (first, end) partial record of each camera (e0_lpl), else nullptr
  const int* cam_hot;      // [n_cams] 1 + rank in the LDS cache, 0 = not cached
  int n_hot_acc, n_hot_wg;
  // state
  double4* cams4;      // [n_cams][3]
  double4* cams_lin4;  // cameras at the linearisation point
  double4* lms4;       // [n_lms] (x, y, z, 1)
  double4* lms_lin4;
  // per landmark
  double4* jl_scale4;  // (s0, s1, s2, -)
  double* hll_inv;     // [n_lms][9]
  double* lmrec;       // [n_lms][12] packed (x,y,z, s0,s1,s2, Hi00,Hi01,Hi02,Hi11,Hi12,Hi22)
  // per slot, dynamic
  double* sw;          // sqrt(robust weight)
  double4* rres;       // weighted residual at the linearisation point
  double4* q4;         // (q0, q1, q2, -): transpose-scatter scalars per slot
  // LDSACC modes: the scalars of the COLD observations go straight to their position in the cold
  // camera-major view (q4c[cold_pos[slot]]), so the per-camera sums stream them instead of gathering
  double4* q4c;        // [n_cold] or nullptr (every other mode: q4[slot])
  const int* cold_pos; // [n_slots] position in the cold view, -1 for observations accumulated in LDS
  int long_in_kernel;  // 1: e0_lm_cached<true> walks the long landmarks itself (dealt round-robin to its wavefronts)
  double2* tiles;      // stored-tile mode: [n_bins][TILE_PAIRS][64] double2
  // per camera
  double* sigma;       // pose_jacobian_scaling [n_cams][12]
  double* diag2;
  double* G;           // [n_cams][40]
  double* binv;        // [n_cams][144]
  double* b;           // [n_cams][12]
  double* tmp;         // current series term
  double* accum;
  double* z;           // sigma * tmp (implicit mode input of E0)
  double* y;           // dense E0 output (tile mode / after all-reduce)
  double* inc;         // pose increment handed to apply
  double* item_part;   // [n_items][12]
  double* item_partG;  // [n_items][40]
  // peer-to-peer exchange of the per-term E0 partials (povar_p2p_attach): every rank pushes its
  // per-camera sums into every peer's exchange buffer and reduces the world's slabs itself, instead of an all-reduce
  double* const* p2p_peer;       // [world] base of each rank's exchange buffer [2][world][n_cams][16] (peer-mapped)
  unsigned long long* p2p_epoch; // device counter, +1 per term (by e0_lpl); tags the slabs
  int p2p_world, p2p_rank;
  // control
  int* flags;          // [0] non-finite seen (bit 1: p2p wait timed out), [1] series done, [2] iterations, [3] status
  double* norm_part;   // [n_cam_blocks][2]
  double* norms;       // [0] norm_0, [1] last term norm, [2] accum norm
  // scalars
  double sa, sb, eps, huber, lambda_lm;
  int robust;
  int scale_jl;  // 1: scale_Jl_cols_pOSE after linearisation (power linearizor), 0: LinearizorSC (no scaling)
};

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

// Eigen fixed-size 3x3 inverse: cofactors / determinant (landmark_block.hpp:518)
__device__ inline void inv3(const double (&m)[9], double (&r)[9]) {
  const double k00 = m[4] * m[8] - m[5] * m[7];
  const double k10 = m[2] * m[7] - m[1] * m[8];
  const double k20 = m[1] * m[5] - m[2] * m[4];
  const double det = k00 * m[0] + k10 * m[3] + k20 * m[6];
  const double id = 1.0 / det;
  r[0] = k00 * id;
  r[1] = k10 * id;
  r[2] = k20 * id;
  r[3] = (m[5] * m[6] - m[3] * m[8]) * id;
  r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r[6] = (m[3] * m[7] - m[4] * m[6]) * id;
  r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// symmetric 3x3 from the 6 accumulated upper entries (00,01,02,11,12,22)
__device__ inline void sym3(const double* u, double (&m)[9]) {
  m[0] = u[0]; m[1] = u[1]; m[2] = u[2];
  m[3] = u[1]; m[4] = u[3]; m[5] = u[4];
  m[6] = u[2]; m[7] = u[4]; m[8] = u[5];
}

// compute_error_weight, bal_bundle_adjustment_helper.cpp:52-74
__device__ inline void error_weight(const Dp& d, double r2, double& e, double& w) {
  if (d.robust == 1) {
    const double t = d.huber;
    w = r2 < t * t ? 1.0 : t / sqrt(r2);
    e = 0.5 * (2 - w) * w * r2;
  } else if (d.robust == 2) {
    w = 1.0;
    e = log(1.0 + r2);
  } else {
    w = 1.0;
    e = 0.5 * r2;
  }
}

struct Cam {
  double4 r0, r1, r2;
};
__device__ inline Cam load_cam(const double4* cams4, int c) {
  Cam P;
  P.r0 = cams4[3 * c];
  P.r1 = cams4[3 * c + 1];
  P.r2 = cams4[3 * c + 2];
  return P;
}
__device__ inline double dot4(const double4& a, const double4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// pOSE residual (bal_bundle_adjustment_helper.cpp:250-261): M = [sb(P0-uP2); sb(P1-vP2); saP0; saP1]
__device__ inline void pose_residual(const Dp& d, const Cam& P, const double4& h, double u, double v,
                                     double (&res)[4]) {
  double4 m0, m1, m2, m3;
  m0.x = d.sb * (P.r0.x - P.r2.x * u); m0.y = d.sb * (P.r0.y - P.r2.y * u);
  m0.z = d.sb * (P.r0.z - P.r2.z * u); m0.w = d.sb * (P.r0.w - P.r2.w * u);
  m1.x = d.sb * (P.r1.x - P.r2.x * v); m1.y = d.sb * (P.r1.y - P.r2.y * v);
  m1.z = d.sb * (P.r1.z - P.r2.z * v); m1.w = d.sb * (P.r1.w - P.r2.w * v);
  m2.x = d.sa * P.r0.x; m2.y = d.sa * P.r0.y; m2.z = d.sa * P.r0.z; m2.w = d.sa * P.r0.w;
  m3.x = d.sa * P.r1.x; m3.y = d.sa * P.r1.y; m3.z = d.sa * P.r1.z; m3.w = d.sa * P.r1.w;
  res[0] = dot4(m0, h);
  res[1] = dot4(m1, h);
  res[2] = dot4(m2, h) - d.sa * u;
  res[3] = dot4(m3, h) - d.sa * v;
}

// Jl = scale * M[:, :3] * diag(s)  (row-major 4x3); scale = sqrt(w), s = Jl column scale
__device__ inline void pose_jl(const Dp& d, const Cam& P, double u, double v, double scale,
                               const double4& s, double (&jl)[12]) {
  const double cb = d.sb * scale, ca = d.sa * scale;
  jl[0] = cb * (P.r0.x - P.r2.x * u) * s.x;
  jl[1] = cb * (P.r0.y - P.r2.y * u) * s.y;
  jl[2] = cb * (P.r0.z - P.r2.z * u) * s.z;
  jl[3] = cb * (P.r1.x - P.r2.x * v) * s.x;
  jl[4] = cb * (P.r1.y - P.r2.y * v) * s.y;
  jl[5] = cb * (P.r1.z - P.r2.z * v) * s.z;
  jl[6] = ca * P.r0.x * s.x;
  jl[7] = ca * P.r0.y * s.y;
  jl[8] = ca * P.r0.z * s.z;
  jl[9] = ca * P.r1.x * s.x;
  jl[10] = ca * P.r1.y * s.y;
  jl[11] = ca * P.r1.z * s.z;
}

// t = Jp x for the structured Jp, zc = (sigma*x)[12c..12c+12) as three double4
__device__ inline void pose_jp_x(const Dp& d, const double4& h, double u, double v, double scale,
                                 const double4* zc, double (&t)[4]) {
  const double d0 = dot4(h, zc[0]), d1 = dot4(h, zc[1]), d2 = dot4(h, zc[2]);
  t[0] = d.sb * scale * (d0 - u * d2);
  t[1] = d.sb * scale * (d1 - v * d2);
  t[2] = d.sa * scale * d0;
  t[3] = d.sa * scale * d1;
}

// q of Jp^T s = sigma * (h q0; h q1; h q2)
__device__ inline double4 pose_q(const Dp& d, double u, double v, double scale, const double (&s)[4]) {
  double4 q;
  q.x = scale * (d.sb * s[0] + d.sa * s[2]);
  q.y = scale * (d.sb * s[1] + d.sa * s[3]);
  q.z = -scale * d.sb * (u * s[0] + v * s[1]);
  q.w = scale;
  return q;
}

// where the transpose-scatter scalars of one observation go (see Dp::q4c)
__device__ inline void store_q(const Dp& d, int slot, const double4& q) {
  if (d.q4c) d.q4c[d.cold_pos[slot]] = q;
  else d.q4[slot] = q;
}

__device__ inline double shfl_up_d(double v, int delta) { return __shfl_up(v, delta, WAVE); }
__device__ inline double shfl_d(double v, int src) { return __shfl(v, src, WAVE); }
__device__ inline double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, WAVE); }


// Per-landmark sums: inclusive segmented scan inside a wavefront, then broadcast of the segment total.
// `steps` bounds the in-row doubling steps by the longest segment of the wavefront (wave-uniform; landmarks
// average 4-6 observations, so 3 steps instead of 6).  The scan runs on the VALU's DPP network instead of
// ds_bpermute round trips through the LDS crossbar: in-row Hillis-Steele steps (row_shr 1/2/4/8, out-of-row
// sources read 0), then the carries across the three 16-lane row boundaries one row at a time
// (row_bcast:15 under a row mask), then the broadcast of the segment totals -- the only ds_bpermute left
// (e0_lm_cached<true>: 106.9 -> 101.0 us).  The association of a segment's sum depends on where the row
// boundaries cut it: results are bit-reproducible for a given layout, not across landmark orders.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_dm(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
// (body shared with seg_scan_steps: BCAST = false leaves the inclusive scan -- the segment's total is in its LAST lane --
// and saves the 2 N ds_bpermute of the broadcast: what the callers that only write the total from one lane want)
template <int N, bool BCAST>
__device__ inline void seg_scan_impl(double (&v)[N], int lane, int seg_first, int seg_last, int steps);
template <int N>
__device__ inline void seg_reduce_steps(double (&v)[N], int lane, int seg_first, int seg_last, int steps) {
  seg_scan_impl<N, true>(v, lane, seg_first, seg_last, steps);
}
template <int N>
__device__ inline void seg_scan_steps(double (&v)[N], int lane, int seg_first, int steps) {
  seg_scan_impl<N, false>(v, lane, seg_first, 0, steps);
}
template <int N, bool BCAST>
__device__ inline void seg_scan_impl(double (&v)[N], int lane, int seg_first, int seg_last, int steps) {
  auto mask = [](bool c) { return __hiloint2double(c ? 0x3FF00000 : 0, 0); };
  const double m1 = mask(lane - 1 >= seg_first), m2 = mask(lane - 2 >= seg_first), m4 = mask(lane - 4 >= seg_first),
               m8 = mask(lane - 8 >= seg_first), mc = mask(seg_first < (lane & ~15));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fma(m1, dpp_dm<0x111, 0xf>(v[k]), v[k]);
  if (steps > 1) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fma(m2, dpp_dm<0x112, 0xf>(v[k]), v[k]);
  }
  if (steps > 2) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fma(m4, dpp_dm<0x114, 0xf>(v[k]), v[k]);
  }
  if (steps > 3) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fma(m8, dpp_dm<0x118, 0xf>(v[k]), v[k]);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fma(mc, dpp_dm<0x142, 0x2>(v[k]), v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fma(mc, dpp_dm<0x142, 0x4>(v[k]), v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fma(mc, dpp_dm<0x142, 0x8>(v[k]), v[k]);
  if (BCAST) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = shfl_d(v[k], seg_last);
  }
}

// value of lane (src lane per the DPP control), 0 where the source lane does not exist (bound_ctrl)
template <int CTRL>
__device__ inline double dpp_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// wavefront sum through the VALU's DPP network (row_shr 1/2/4/8, row_bcast 15/31): the totals end up in
// lane 63 only.  Six dependent v_mov_dpp + v_add_f64 steps instead of six ds_bpermute round trips through
// the LDS crossbar -- the per-camera kernels are one latency chain, so this is their critical path.
template <int N>
__device__ inline void wave_total_dpp(double (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_d<0x111>(v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_d<0x112>(v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_d<0x114>(v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_d<0x118>(v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_d<0x142>(v[k]);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_d<0x143>(v[k]);
}
// value of `v` in a compile-time lane, through v_readlane (SGPR broadcast: no LDS crossbar, no VGPRs)
__device__ inline double bcast_lane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// wavefront sum, total in every lane (DPP reduction to lane 63, then a scalar broadcast)
template <int N>
__device__ inline void wave_sum(double (&v)[N]) {
  wave_total_dpp<N>(v);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = bcast_lane(v[k], 63);
}

// deterministic workgroup sum of N per-thread values -> out[N] valid in every thread
template <int N, int BLOCK>
__device__ inline void block_sum(double (&v)[N], double* sh /* [BLOCK/64][N] */) {
  wave_sum<N>(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) sh[w * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double s = 0;
    for (int i = 0; i < BLOCK / 64; ++i) s += sh[i * N + k];
    v[k] = s;
  }
}

// deterministic workgroup sum like block_sum, wavefront stage on DPP
template <int N, int BLOCK>
__device__ inline void block_sum_dpp(double (&v)[N], double* sh /* [BLOCK/64][N] */) {
  wave_total_dpp<N>(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < N; ++k) sh[w * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double s = 0;
    for (int i = 0; i < BLOCK / 64; ++i) s += sh[i * N + k];
    v[k] = s;
  }
}

// ------------------------------------------------------------------------------------------
// landmark-major drivers
// ------------------------------------------------------------------------------------------
//
// An Op provides
//   NRED  per-landmark reduction width, NSC  global scalar reduction width
//   Local per-observation registers carried from phase1 to phase2
//   phase1(d, slot, cam, lm, uv, L, red)            per observation: inputs -> partial sums
//   phase2(d, slot, cam, lm, uv, L, tot, sc)        per observation: totals -> outputs
//   finish_lm(d, lm, tot)                           once per landmark

template <class Op>
__global__ __launch_bounds__(LM_BLOCK) void lm_regular(Dp d, Op op, double* part) {
  if (Op::CHECK_DONE && d.flags[1]) return;
  const int slot = blockIdx.x * LM_BLOCK + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = slot < d.n_bins * WAVE;
  const int meta = in ? d.meta[slot] : (lane | (lane << 8));
  const bool valid = in && (meta & META_REAL) && !(meta & META_LONG);
  const int seg_first = meta & 255, seg_last = (meta >> 8) & 255;
  int cam = 0, lm = 0;
  double2 uv = make_double2(0, 0);
  if (valid) {
    cam = d.cam[slot];
    lm = d.lm[slot];
    uv = d.uv[slot];
  }
  typename Op::Local L;
  double red[Op::NRED > 0 ? Op::NRED : 1];
#pragma unroll
  for (int k = 0; k < (Op::NRED > 0 ? Op::NRED : 1); ++k) red[k] = 0;
  if (valid) op.phase1(d, slot, cam, lm, uv, L, red);
  if constexpr (Op::NRED > 0)
    seg_reduce_steps<Op::NRED>(red, lane, seg_first, seg_last,
                               __builtin_amdgcn_readfirstlane((meta >> META_STEPS_SHIFT) & 7));
  double sc[Op::NSC > 0 ? Op::NSC : 1];
#pragma unroll
  for (int k = 0; k < (Op::NSC > 0 ? Op::NSC : 1); ++k) sc[k] = 0;
  if (valid) {
    op.phase2(d, slot, cam, lm, uv, L, red, sc);
    if (lane == seg_last) op.finish_lm(d, lm, red);
  }
  if constexpr (Op::NSC > 0) {
    __shared__ double sh[(LM_BLOCK / 64) * Op::NSC];
    block_sum<Op::NSC, LM_BLOCK>(sc, sh);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < Op::NSC; ++k) part[(size_t)blockIdx.x * Op::NSC + k] = sc[k];
    }
  }
}

template <class Op>
__global__ __launch_bounds__(LM_BLOCK) void lm_long(Dp d, Op op, double* part) {
  if (Op::CHECK_DONE && d.flags[1]) return;
  const int lm = d.long_lm[blockIdx.x];
  const int first = d.long_first[blockIdx.x];
  const int cnt = d.long_cnt[blockIdx.x];
  constexpr int NR = Op::NRED > 0 ? Op::NRED : 1;
  constexpr int NS = Op::NSC > 0 ? Op::NSC : 1;
  __shared__ double sh[(LM_BLOCK / 64) * (NR > NS ? NR : NS)];
  double tot[NR];
#pragma unroll
  for (int k = 0; k < NR; ++k) tot[k] = 0;
  double sc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sc[k] = 0;
  if (cnt <= LM_BLOCK) {
    // one observation per thread: the phase-1 state stays in registers across the landmark sum
    const bool in = (int)threadIdx.x < cnt;
    const int slot = first + (in ? (int)threadIdx.x : 0);
    const int cam = d.cam[slot];
    const double2 uv = d.uv[slot];
    typename Op::Local L;
    if (in) op.phase1(d, slot, cam, lm, uv, L, tot);
    if constexpr (Op::NRED > 0) block_sum<NR, LM_BLOCK>(tot, sh);
    if (in) op.phase2(d, slot, cam, lm, uv, L, tot, sc);
  } else {
    for (int i = threadIdx.x; i < cnt; i += LM_BLOCK) {
      const int slot = first + i;
      typename Op::Local L;
      double red[NR];
#pragma unroll
      for (int k = 0; k < NR; ++k) red[k] = 0;
      op.phase1(d, slot, d.cam[slot], lm, d.uv[slot], L, red);
#pragma unroll
      for (int k = 0; k < NR; ++k) tot[k] += red[k];
    }
    if constexpr (Op::NRED > 0) block_sum<NR, LM_BLOCK>(tot, sh);
    for (int i = threadIdx.x; i < cnt; i += LM_BLOCK) {  // longer landmarks: phase 1 recomputed
      const int slot = first + i;
      typename Op::Local L;
      double red[NR];
#pragma unroll
      for (int k = 0; k < NR; ++k) red[k] = 0;
      op.phase1(d, slot, d.cam[slot], lm, d.uv[slot], L, red);
      op.phase2(d, slot, d.cam[slot], lm, d.uv[slot], L, tot, sc);
    }
  }
  __syncthreads();  // every phase1 read of per-landmark state is done before finish_lm writes it
  if (threadIdx.x == 0) op.finish_lm(d, lm, tot);
  if constexpr (Op::NSC > 0) {
    block_sum<NS, LM_BLOCK>(sc, sh);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) part[(size_t)(d.n_reg_blocks + blockIdx.x) * NS + k] = sc[k];
    }
  }
}

// fixed-order sum of per-workgroup partials: out[k] = sum_i part[i*N + k]
template <int N>
__global__ __launch_bounds__(1024) void reduce_partials(const double* part, int n, double* out) {
  __shared__ double sh[16 * N];
  double v[N];
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = 0;
  for (int i = threadIdx.x; i < n; i += 1024) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += part[(size_t)i * N + k];
  }
  block_sum<N, 1024>(v, sh);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = v[k];
  }
}

// ------------------------------------------------------------------------------------------
// landmark-major ops
// ------------------------------------------------------------------------------------------

struct NoLocal {};

// K1: initialize_varproj_lm_pOSE (bal_bundle_adjustment_helper.cpp:76-99, 221-241).
// x_l = argmin |G x - z|: the reference solves it with bdcSvd; here the 3x3 normal equations
// (G^T G) x = G^T z are accumulated and solved with the closed-form inverse, followed by one
// refinement step (OpInitRefine).
struct OpInit {
  static constexpr int NRED = 9, NSC = 0;
  static constexpr bool CHECK_DONE = false;
  using Local = NoLocal;
  __device__ void phase1(const Dp& d, int, int cam, int, double2 uv, Local&, double* red) const {
    const Cam P = load_cam(d.cams4, cam);
    const double4 one = make_double4(1, 1, 1, 1);
    double g[12];
    pose_jl(d, P, uv.x, uv.y, 1.0, one, g);
    const double z[4] = {d.sb * (P.r2.w * uv.x - P.r0.w), d.sb * (P.r2.w * uv.y - P.r1.w),
                         d.sa * (uv.x - P.r0.w), d.sa * (uv.y - P.r1.w)};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[0] += g[3 * r] * g[3 * r];
      red[1] += g[3 * r] * g[3 * r + 1];
      red[2] += g[3 * r] * g[3 * r + 2];
      red[3] += g[3 * r + 1] * g[3 * r + 1];
      red[4] += g[3 * r + 1] * g[3 * r + 2];
      red[5] += g[3 * r + 2] * g[3 * r + 2];
      red[6] += g[3 * r] * z[r];
      red[7] += g[3 * r + 1] * z[r];
      red[8] += g[3 * r + 2] * z[r];
    }
  }
  __device__ void phase2(const Dp&, int, int, int, double2, Local&, const double*, double*) const {}
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    double H[9], Hi[9];
    sym3(tot, H);
    inv3(H, Hi);
    double4 x;
    x.x = Hi[0] * tot[6] + Hi[1] * tot[7] + Hi[2] * tot[8];
    x.y = Hi[3] * tot[6] + Hi[4] * tot[7] + Hi[5] * tot[8];
    x.z = Hi[6] * tot[6] + Hi[7] * tot[7] + Hi[8] * tot[8];
    x.w = 1.0;
    d.lms4[lm] = x;
  }
};

// K1 as a least-squares solve with the conditioning of the reference's bdcSvd().solve (HLP:94): one thread per
// landmark folds the 4 rows of every observation into an upper-triangular 3x4 [R | c] with Givens rotations
// (an incremental QR of G; the error grows with kappa(G), not with kappa(G)^2 as for the normal equations) and
// back-substitutes.  Near-parallel two-view landmarks (kappa 1e6..1e7) keep ~1e-9; the normal-equation kernels
// OpInit / OpInitRefine lose them (tests/test_gpu_fuzz.py::test_init_landmarks_near_degenerate).  Runs once per
// solve (linearizor_base.cpp:61-67), so the uncoalesced per-landmark walk does not matter.
POVAR_KERNEL __launch_bounds__(256) void init_landmarks_qr(Dp d) {
  const int lm = blockIdx.x * 256 + threadIdx.x;
  if (lm >= d.n_lms) return;
  const int s0 = d.lm_slot0[lm], k = d.lm_cnt[lm];
  double R[6] = {0, 0, 0, 0, 0, 0}, c[3] = {0, 0, 0};  // R = [r00 r01 r02; 0 r11 r12; 0 0 r22]
  for (int i = 0; i < k; ++i) {
    const int slot = s0 + i;
    const Cam P = load_cam(d.cams4, d.cam[slot]);
    const double2 uv = d.uv[slot];
    double g[12];
    pose_jl(d, P, uv.x, uv.y, 1.0, make_double4(1, 1, 1, 1), g);
    const double z[4] = {d.sb * (P.r2.w * uv.x - P.r0.w), d.sb * (P.r2.w * uv.y - P.r1.w),
                         d.sa * (uv.x - P.r0.w), d.sa * (uv.y - P.r1.w)};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double a0 = g[3 * r], a1 = g[3 * r + 1], a2 = g[3 * r + 2], b = z[r];
      // eliminate a0 against r00, a1 against r11, a2 against r22
      {
        const double rho = hypot(R[0], a0);
        if (rho > 0) {
          const double cs = R[0] / rho, sn = a0 / rho;
          R[0] = rho;
          double t = cs * R[1] + sn * a1; a1 = cs * a1 - sn * R[1]; R[1] = t;
          t = cs * R[2] + sn * a2; a2 = cs * a2 - sn * R[2]; R[2] = t;
          t = cs * c[0] + sn * b; b = cs * b - sn * c[0]; c[0] = t;
        }
      }
      {
        const double rho = hypot(R[3], a1);
        if (rho > 0) {
          const double cs = R[3] / rho, sn = a1 / rho;
          R[3] = rho;
          double t = cs * R[4] + sn * a2; a2 = cs * a2 - sn * R[4]; R[4] = t;
          t = cs * c[1] + sn * b; b = cs * b - sn * c[1]; c[1] = t;
        }
      }
      {
        const double rho = hypot(R[5], a2);
        if (rho > 0) {
          const double cs = R[5] / rho, sn = a2 / rho;
          R[5] = rho;
          const double t = cs * c[2] + sn * b;
          c[2] = t;
        }
      }
    }
  }
  double4 x;
  x.z = c[2] / R[5];
  x.y = (c[1] - R[4] * x.z) / R[3];
  x.x = (c[0] - R[1] * x.y - R[2] * x.z) / R[0];
  x.w = 1.0;
  if (k > 0) d.lms4[lm] = x;
}

// One step of iterative refinement for K1: x += (G^T G)^-1 G^T (z - G x) with the residual taken
// per observation at the current x.  The normal equations square the condition number of G; the
// refinement step brings the result back to the accuracy of the reference's SVD solve for the
// moderately ill-conditioned two- and three-view landmarks (error ~ (kappa^2 u)^2 instead of kappa^2 u).
struct OpInitRefine {
  static constexpr int NRED = 9, NSC = 0;
  static constexpr bool CHECK_DONE = false;
  using Local = NoLocal;
  __device__ void phase1(const Dp& d, int, int cam, int lm, double2 uv, Local&, double* red) const {
    const Cam P = load_cam(d.cams4, cam);
    const double4 one = make_double4(1, 1, 1, 1);
    double g[12];
    pose_jl(d, P, uv.x, uv.y, 1.0, one, g);
    const double4 x = d.lms4[lm];
    const double z[4] = {d.sb * (P.r2.w * uv.x - P.r0.w), d.sb * (P.r2.w * uv.y - P.r1.w),
                         d.sa * (uv.x - P.r0.w), d.sa * (uv.y - P.r1.w)};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double res = z[r] - (g[3 * r] * x.x + g[3 * r + 1] * x.y + g[3 * r + 2] * x.z);
      red[0] += g[3 * r] * g[3 * r];
      red[1] += g[3 * r] * g[3 * r + 1];
      red[2] += g[3 * r] * g[3 * r + 2];
      red[3] += g[3 * r + 1] * g[3 * r + 1];
      red[4] += g[3 * r + 1] * g[3 * r + 2];
      red[5] += g[3 * r + 2] * g[3 * r + 2];
      red[6] += g[3 * r] * res;
      red[7] += g[3 * r + 1] * res;
      red[8] += g[3 * r + 2] * res;
    }
  }
  __device__ void phase2(const Dp&, int, int, int, double2, Local&, const double*, double*) const {}
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    double H[9], Hi[9];
    sym3(tot, H);
    inv3(H, Hi);
    double4 x = d.lms4[lm];
    x.x += Hi[0] * tot[6] + Hi[1] * tot[7] + Hi[2] * tot[8];
    x.y += Hi[3] * tot[6] + Hi[4] * tot[7] + Hi[5] * tot[8];
    x.z += Hi[6] * tot[6] + Hi[7] * tot[7] + Hi[8] * tot[8];
    d.lms4[lm] = x;
  }
};

// K2: compute_error_pOSE (bal_bundle_adjustment_helper.cpp:117-154)
struct OpError {
  static constexpr int NRED = 0, NSC = 3;
  static constexpr bool CHECK_DONE = false;
  using Local = NoLocal;
  __device__ void phase1(const Dp&, int, int, int, double2, Local&, double*) const {}
  __device__ void phase2(const Dp& d, int, int cam, int lm, double2 uv, Local&, const double*,
                         double* sc) const {
    const Cam P = load_cam(d.cams4, cam);
    const double4 h = d.lms4[lm];
    double res[4];
    pose_residual(d, P, h, uv.x, uv.y, res);
    const double r2 = res[0] * res[0] + res[1] * res[1] + res[2] * res[2] + res[3] * res[3];
    if (!isfinite(r2)) atomicOr(&d.flags[0], 1);
    double e, w;
    error_weight(d, r2, e, w);
    sc[0] += e;
    sc[1] += sqrt(r2);
    sc[2] += 1.0;
  }
  __device__ void finish_lm(const Dp&, int, const double*) const {}
};

// K3 + K5: linearize_landmark_pOSE (landmark_block.hpp:135-178) and scale_Jl_cols_pOSE
// (landmark_block.hpp:284-295).  Keeps sqrt(w) and the weighted residual per observation and
// the Jl column scale per landmark; the tiles themselves are implicit.
struct OpLinearize {
  static constexpr int NRED = 3, NSC = 0;
  static constexpr bool CHECK_DONE = false;
  using Local = NoLocal;
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local&, double* red) const {
    const Cam P = load_cam(d.cams_lin4, cam);
    const double4 h = d.lms_lin4[lm];
    double res[4];
    pose_residual(d, P, h, uv.x, uv.y, res);
    const double r2 = res[0] * res[0] + res[1] * res[1] + res[2] * res[2] + res[3] * res[3];
    double e, w;
    error_weight(d, r2, e, w);
    const double sw = sqrt(w);
    if (!isfinite(r2) || !isfinite(sw)) atomicOr(&d.flags[0], 1);
    d.sw[slot] = sw;
    if (d.robust && d.v2.w && !d.lin_aux_only) d.v2.w[d.v2.of_slot[slot]] = w;
    d.rres[slot] = make_double4(sw * res[0], sw * res[1], sw * res[2], sw * res[3]);
    double jl[12];
    pose_jl(d, P, uv.x, uv.y, sw, make_double4(1, 1, 1, 1), jl);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[0] += jl[3 * r] * jl[3 * r];
      red[1] += jl[3 * r + 1] * jl[3 * r + 1];
      red[2] += jl[3 * r + 2] * jl[3 * r + 2];
    }
  }
  __device__ void phase2(const Dp&, int, int, int, double2, Local&, const double*, double*) const {}
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    if (d.lin_aux_only) return;  // only the per-slot sqrt(w) / residual arrays are wanted (ensure_legacy, povar_lm.hip)
    // LinearizorSC::linearize_pOSE leaves the Jl columns unscaled (linearizor_sc.cpp:163-191)
    d.jl_scale4[lm] = d.scale_jl ? make_double4(1.0 / (d.eps + sqrt(tot[0])), 1.0 / (d.eps + sqrt(tot[1])),
                                                1.0 / (d.eps + sqrt(tot[2])), 0.0)
                                 : make_double4(1.0, 1.0, 1.0, 0.0);
  }
};

// K7 (landmark part): get_Hll_inv_add_Hpp_b_pOSE / _poBA (landmark_block.hpp:510-572).
// Hll = Jl^T Jl (+ lambda I for POWER_SCHUR_COMPLEMENT), Hll^-1, w = Hll^-1 Jl^T r and the
// scatter scalars of Jp^T (r - Jl w); the per-camera sums are taken by cm_scatter.
struct OpPrepare {
  static constexpr int NRED = 9, NSC = 0;
  static constexpr bool CHECK_DONE = false;
  struct Local {
    double jl[12];
    double4 r;
    double sw;
  };
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local& L, double* red) const {
    const Cam P = load_cam(d.cams_lin4, cam);
    L.sw = d.robust ? d.sw[slot] : 1.0;
    L.r = d.rres[slot];
    pose_jl(d, P, uv.x, uv.y, L.sw, d.jl_scale4[lm], L.jl);
    const double rr[4] = {L.r.x, L.r.y, L.r.z, L.r.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[0] += L.jl[3 * r] * L.jl[3 * r];
      red[1] += L.jl[3 * r] * L.jl[3 * r + 1];
      red[2] += L.jl[3 * r] * L.jl[3 * r + 2];
      red[3] += L.jl[3 * r + 1] * L.jl[3 * r + 1];
      red[4] += L.jl[3 * r + 1] * L.jl[3 * r + 2];
      red[5] += L.jl[3 * r + 2] * L.jl[3 * r + 2];
      red[6] += L.jl[3 * r] * rr[r];
      red[7] += L.jl[3 * r + 1] * rr[r];
      red[8] += L.jl[3 * r + 2] * rr[r];
    }
  }
  __device__ static void hinv(const Dp& d, const double* tot, double (&Hi)[9]) {
    double H[9];
    sym3(tot, H);
    H[0] += d.lambda_lm;
    H[4] += d.lambda_lm;
    H[8] += d.lambda_lm;
    inv3(H, Hi);
  }
  __device__ void phase2(const Dp& d, int slot, int, int, double2 uv, Local& L, const double* tot,
                         double*) const {
    double Hi[9];
    hinv(d, tot, Hi);
    const double w0 = Hi[0] * tot[6] + Hi[1] * tot[7] + Hi[2] * tot[8];
    const double w1 = Hi[3] * tot[6] + Hi[4] * tot[7] + Hi[5] * tot[8];
    const double w2 = Hi[6] * tot[6] + Hi[7] * tot[7] + Hi[8] * tot[8];
    const double rr[4] = {L.r.x, L.r.y, L.r.z, L.r.w};
    double e[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      e[r] = rr[r] - (L.jl[3 * r] * w0 + L.jl[3 * r + 1] * w1 + L.jl[3 * r + 2] * w2);
    d.q4[slot] = pose_q(d, uv.x, uv.y, L.sw, e);
  }
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    double Hi[9];
    hinv(d, tot, Hi);
#pragma unroll
    for (int k = 0; k < 9; ++k) d.hll_inv[9 * (size_t)lm + k] = Hi[k];
    // packed per-landmark record for the per-term kernels (Hll^-1 is exactly symmetric)
    const double4 h = d.lms_lin4[lm], s = d.jl_scale4[lm];
    double4* rec = reinterpret_cast<double4*>(d.lmrec) + 3 * (size_t)lm;
    rec[0] = make_double4(h.x, h.y, h.z, s.x);
    rec[1] = make_double4(s.y, s.z, Hi[0], Hi[1]);
    rec[2] = make_double4(Hi[2], Hi[4], Hi[5], Hi[8]);
    if (d.v2.lmrec && !d.prep_aux_only) {
      // record of the lane-per-landmark kernel: the Jl column scale is folded into G = diag(s) Hll^-1 diag(s)
      const int lp = d.v2.lm_pos[lm], pos = lp & ((1 << 26) - 1), lanes = ((lp >> 26) & 63) + 1;
      const double G[6] = {s.x * Hi[0] * s.x, s.x * Hi[1] * s.y, s.x * Hi[2] * s.z,
                           s.y * Hi[4] * s.y, s.y * Hi[5] * s.z, s.z * Hi[8] * s.z};
      for (int q = 0; q < lanes; ++q) {  // one copy per lane the landmark occupies (adjacent, same tile)
        double* r2 = d.v2.lmrec + ((size_t)(pos >> 6) * 9) * WAVE + (pos & 63) + q;
        r2[0] = h.x; r2[WAVE] = h.y; r2[2 * WAVE] = h.z;
#pragma unroll
        for (int m = 0; m < 6; ++m) r2[(3 + m) * WAVE] = G[m];
      }
    }
  }
};

// K10 (implicit, landmark part): right_mul_e0_pOSE (linearization_power_varproj.hpp:364-406).
// t = Jp x, u = Jl^T t (segmented sum over the landmark), v = Hll^-1 u, s = Jl v, then the three
// scatter scalars of Jp^T s.  Input z = sigma * x.
// Algebra used here (exact, with sa^2 + sb^2 = 1): with d_k = h . z_c[4k..4k+4], P3 = P_c[:, :3],
// s = Jl column scale, w = sqrt-weight^2 and the 3x3 matrix of the pOSE residual
//     C(u,v) = [[1, 0, -sb^2 u], [0, 1, -sb^2 v], [-sb^2 u, -sb^2 v, sb^2 (u^2 + v^2)]]
// the four rows of Jl and Jp collapse:   Jl^T (Jp x) = s * (P3^T (w C d)),   Jp^T (Jl v) -> q = w C (P3 (s * v)).
// 73 fp64 operations per observation instead of 110, and only P3 (not the 4x3 Jl) lives across the
// segmented scan.
struct E0Core {
  double P3[9];
  double w, cu, cv, cuv;  // w, sb^2 u, sb^2 v, sb^2 (u^2 + v^2)
  __device__ inline void forward(const Dp& d, const double* P3_, const double* zz, const double4& rec0,
                                 const double4& rec1, double2 uv, double sw, double* red) {
#pragma unroll
    for (int i = 0; i < 9; ++i) P3[i] = P3_[i];
    const double sb2 = d.sb * d.sb;
    w = sw * sw;
    cu = sb2 * uv.x;
    cv = sb2 * uv.y;
    cuv = sb2 * (uv.x * uv.x + uv.y * uv.y);
    const double hx = rec0.x, hy = rec0.y, hz = rec0.z;
    const double d0 = hx * zz[0] + hy * zz[1] + hz * zz[2] + zz[3];
    const double d1 = hx * zz[4] + hy * zz[5] + hz * zz[6] + zz[7];
    const double d2 = hx * zz[8] + hy * zz[9] + hz * zz[10] + zz[11];
    const double a0 = w * (d0 - cu * d2);
    const double a1 = w * (d1 - cv * d2);
    const double a2 = w * (cuv * d2 - cu * d0 - cv * d1);
    red[0] += rec0.w * (P3[0] * a0 + P3[3] * a1 + P3[6] * a2);
    red[1] += rec1.x * (P3[1] * a0 + P3[4] * a1 + P3[7] * a2);
    red[2] += rec1.y * (P3[2] * a0 + P3[5] * a1 + P3[8] * a2);
  }
  __device__ inline double4 backward(const Dp&, const double4& rec0, const double4& rec1, const double4& rec2,
                                     const double* tot) const {
    const double h00 = rec1.z, h01 = rec1.w, h02 = rec2.x, h11 = rec2.y, h12 = rec2.z, h22 = rec2.w;
    const double g0 = rec0.w * (h00 * tot[0] + h01 * tot[1] + h02 * tot[2]);
    const double g1 = rec1.x * (h01 * tot[0] + h11 * tot[1] + h12 * tot[2]);
    const double g2 = rec1.y * (h02 * tot[0] + h12 * tot[1] + h22 * tot[2]);
    const double e0 = P3[0] * g0 + P3[1] * g1 + P3[2] * g2;
    const double e1 = P3[3] * g0 + P3[4] * g1 + P3[5] * g2;
    const double e2 = P3[6] * g0 + P3[7] * g1 + P3[8] * g2;
    double4 q;
    q.x = w * (e0 - cu * e2);
    q.y = w * (e1 - cv * e2);
    q.z = w * (cuv * e2 - cu * e0 - cv * e1);
    q.w = 0;  // (unused slot of q4)
    return q;
  }
};

__device__ inline void load_cam_global(const Dp& d, int cam, double* P3, double* zz) {
  const Cam P = load_cam(d.cams_lin4, cam);
  P3[0] = P.r0.x; P3[1] = P.r0.y; P3[2] = P.r0.z;
  P3[3] = P.r1.x; P3[4] = P.r1.y; P3[5] = P.r1.z;
  P3[6] = P.r2.x; P3[7] = P.r2.y; P3[8] = P.r2.z;
  const double4* zc = reinterpret_cast<const double4*>(d.z) + 3 * cam;
  const double4 z0 = zc[0], z1 = zc[1], z2 = zc[2];
  zz[0] = z0.x; zz[1] = z0.y; zz[2] = z0.z; zz[3] = z0.w;
  zz[4] = z1.x; zz[5] = z1.y; zz[6] = z1.z; zz[7] = z1.w;
  zz[8] = z2.x; zz[9] = z2.y; zz[10] = z2.z; zz[11] = z2.w;
}

// Op form (used by the lm_long driver for landmarks with more than 64 observations)
struct OpE0 {
  static constexpr int NRED = 3, NSC = 0;
  static constexpr bool CHECK_DONE = true;
  struct Local {
    E0Core core;
    double4 rec0, rec1, rec2;
  };
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local& L, double* red) const {
    double P3[9], zz[12];
    load_cam_global(d, cam, P3, zz);
    const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 3 * (size_t)lm;
    L.rec0 = rec[0];
    L.rec1 = rec[1];
    L.rec2 = rec[2];
    L.core.forward(d, P3, zz, L.rec0, L.rec1, uv, d.robust ? d.sw[slot] : 1.0, red);
  }
  __device__ void phase2(const Dp& d, int slot, int, int, double2 uv, Local& L, const double* tot,
                         double*) const {
    store_q(d, slot, L.core.backward(d, L.rec0, L.rec1, L.rec2, tot));
  }
  __device__ void finish_lm(const Dp&, int, const double*) const {}
};

// The per-term landmark-major E0 kernel.  One 1024-thread workgroup per CU walks a contiguous
// range of wave bins.  The divergent per-observation camera gather (z_c and P_c[:, :3], 168 B,
// 12 x 16-B loads that hit 64 different lines per wave instruction) was the bottleneck of the
// plain version (162 us of venice-1778's term, 70 us with the gather made uniform,
// profiles/r01_b_*): the records of the HOT_MAX most observed cameras are therefore staged in LDS
// once per launch (<= 157 KiB, read back from L2) and only observations of colder cameras gather
// from global memory.
template <bool ACC>
__global__ __launch_bounds__(E0C_BLOCK) void e0_lm_cached(Dp d, int bins_per_wg, double* hot_out) {
  // the series-done flag is requested first but only tested after the LDS staging below: the staging has
  // no global side effects, so the flag's round trip overlaps with it instead of preceding every launch
  const int done = d.flags[1];
  extern __shared__ double2 hot[];  // [n_hot][HOT_REC] (+ [n_hot][12] doubles of accumulators if ACC)
  const int n_hot = ACC ? d.n_hot_acc : d.n_hot;
  double* acc = reinterpret_cast<double*>(hot + n_hot * HOT_REC);
  if (ACC)
    for (int i = threadIdx.x; i < n_hot * 12; i += E0C_BLOCK) acc[i] = 0;
  {
    // coalesced copy of the record image (first HOT_REC double2 of each HOT_REC_STRIDE-double record)
    const double2* src = reinterpret_cast<const double2*>(d.hot_rec);
    for (int i = threadIdx.x; i < n_hot * HOT_REC; i += E0C_BLOCK) {
      const int r = i / HOT_REC, j = i - r * HOT_REC;
      hot[i] = src[r * (HOT_REC_STRIDE / 2) + j];
    }
  }
  __syncthreads();
  if (done) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bin0 = blockIdx.x * bins_per_wg;
  const int bin1 = min(bin0 + bins_per_wg, d.n_bins);
  constexpr int STRIDE = E0C_BLOCK / WAVE;
  // The loop is latency-bound (72 % of the wave cycles parked in s_waitcnt at 4 waves per SIMD,
  // profiles/r01_c_sq_counters.txt): a bin needs two dependent memory round trips (slot data, then
  // the landmark record it points to).  The slot data of the NEXT bin is therefore requested
  // before the current bin's record is consumed.
  // slot data two bins ahead (m_*), one bin ahead (n_*); landmark record one bin ahead (n_rec*)
  int n_meta = lane | (lane << 8), n_cam = 0, n_lm = 0, m_meta = n_meta, m_cam = 0, m_lm = 0;
  int n_cpos = 0, m_cpos = 0;  // ACC: position of the slot in the cold camera-major view (travels with the slot data)
  double2 n_uv = make_double2(0, 0), m_uv = n_uv;
  if (bin0 + wave < bin1) {
    const int s = (bin0 + wave) * WAVE + lane;
    n_meta = d.meta[s]; n_cam = d.cam[s]; n_lm = d.lm[s]; n_uv = d.uv[s];
    if (ACC) n_cpos = d.cold_pos[s];
  }
  if (bin0 + wave + STRIDE < bin1) {
    const int s = (bin0 + wave + STRIDE) * WAVE + lane;
    m_meta = d.meta[s]; m_cam = d.cam[s]; m_lm = d.lm[s]; m_uv = d.uv[s];
    if (ACC) m_cpos = d.cold_pos[s];
  }
  double4 n_rec0, n_rec1, n_rec2;
  {
    const bool v0 = (n_meta & META_REAL) && !(n_meta & META_LONG);
    const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 3 * (size_t)(v0 ? n_lm : 0);
    n_rec0 = rec[0]; n_rec1 = rec[1]; n_rec2 = rec[2];
  }
  for (int bin = bin0 + wave; bin < bin1; bin += STRIDE) {
    const int slot = bin * WAVE + lane;
    const int meta = n_meta, cam = n_cam, lm = n_lm, cpos = n_cpos;
    const double2 uv = n_uv;
    const bool valid = (meta & META_REAL) && !(meta & META_LONG);
    const int seg_first = meta & 255, seg_last = (meta >> 8) & 255;
    const double4 rec0 = n_rec0, rec1 = n_rec1, rec2 = n_rec2;
    (void)lm;
    // rotate the slot pipeline and request the slot data two bins ahead
    n_meta = m_meta; n_cam = m_cam; n_lm = m_lm; n_uv = m_uv; n_cpos = m_cpos;
    if (bin + 2 * STRIDE < bin1) {
      const int s = slot + 2 * STRIDE * WAVE;
      m_meta = d.meta[s]; m_cam = d.cam[s]; m_lm = d.lm[s]; m_uv = d.uv[s];
      if (ACC) m_cpos = d.cold_pos[s];
    } else {
      m_meta = lane | (lane << 8);
    }
    double red[3] = {0, 0, 0};
    E0Core core;
    if (valid) {
      double P3[9], zz[12];
      const int hr = ((meta >> META_HOT_SHIFT) & META_HOT_MASK);
      if (hr > 0 && hr <= n_hot) {
        const double2* h = hot + (hr - 1) * HOT_REC;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const double2 v = h[j];
          zz[2 * j] = v.x;
          zz[2 * j + 1] = v.y;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double2 v = h[6 + j];
          P3[2 * j] = v.x;
          P3[2 * j + 1] = v.y;
        }
        P3[8] = h[10].x;
      } else {
        load_cam_global(d, cam, P3, zz);
      }
      core.forward(d, P3, zz, rec0, rec1, uv, d.robust ? d.sw[slot] : 1.0, red);
    }
    if (bin + STRIDE < bin1) {
      const bool vn = (n_meta & META_REAL) && !(n_meta & META_LONG);
      const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 3 * (size_t)(vn ? n_lm : 0);
      n_rec0 = rec[0]; n_rec1 = rec[1]; n_rec2 = rec[2];
    }
    seg_reduce_steps<3>(red, lane, seg_first, seg_last,
                        __builtin_amdgcn_readfirstlane((meta >> META_STEPS_SHIFT) & 7));
    if (valid) {
      const double4 q = core.backward(d, rec0, rec1, rec2, red);
      const int hr = ((meta >> META_HOT_SHIFT) & META_HOT_MASK);
      if (ACC && hr > 0 && hr <= n_hot) {
        // Jp^T s of a cached camera goes straight into the workgroup's LDS accumulator
        // (ds_add_f64, order not fixed); only colder cameras go through q4 + cm_scatter
        double* a = acc + (hr - 1);  // acc[j][camera]: consecutive cameras on consecutive banks
        const double hx = rec0.x, hy = rec0.y, hz = rec0.z;
        const double v[12] = {hx * q.x, hy * q.x, hz * q.x, q.x, hx * q.y, hy * q.y,
                              hz * q.y, q.y, hx * q.z, hy * q.z, hz * q.z, q.z};
#pragma unroll
        for (int j = 0; j < 12; ++j) __hip_atomic_fetch_add(a + j * n_hot, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (ACC) {
        d.q4c[cpos] = q;  // cold camera: straight to its place in the cold camera-major view
      } else {
        d.q4[slot] = q;
      }
    }
  }
  if (ACC && d.long_in_kernel) {
    // Landmarks with more than 64 observations: one wavefront walks the landmark's slots twice (forward sums,
    // then the backward pass) with the same LDS camera cache and accumulators as the bins above, instead of the
    // generic lm_long kernel (cameras from L2, every observation through the cold scatter path).
    // (dealt round-robin over all wavefronts of the grid: tracks cluster in landmark order, workgroups must not)
    for (int j = blockIdx.x * STRIDE + wave; j < d.n_long; j += gridDim.x * STRIDE) {
      const int lm = d.long_lm[j], first = d.long_first[j], cnt = d.long_cnt[j];
      const double4* rec = reinterpret_cast<const double4*>(d.lmrec) + 3 * (size_t)lm;
      const double4 rec0 = rec[0], rec1 = rec[1], rec2 = rec[2];
      double tot[3] = {0, 0, 0};
      for (int pass = 0; pass < 2; ++pass) {
        for (int i0 = 0; i0 < cnt; i0 += WAVE) {
          const bool in = i0 + lane < cnt;
          const int slot = first + (in ? i0 + lane : 0);
          double red[3] = {0, 0, 0};
          E0Core core;
          int hr = 0;
          if (in) {
            const int meta = d.meta[slot];
            hr = (meta >> META_HOT_SHIFT) & META_HOT_MASK;
            double P3[9], zz[12];
            if (hr > 0 && hr <= n_hot) {
              const double2* h = hot + (hr - 1) * HOT_REC;
#pragma unroll
              for (int k = 0; k < 6; ++k) {
                const double2 v = h[k];
                zz[2 * k] = v.x;
                zz[2 * k + 1] = v.y;
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const double2 v = h[6 + k];
                P3[2 * k] = v.x;
                P3[2 * k + 1] = v.y;
              }
              P3[8] = h[10].x;
            } else {
              load_cam_global(d, d.cam[slot], P3, zz);
            }
            core.forward(d, P3, zz, rec0, rec1, d.uv[slot], d.robust ? d.sw[slot] : 1.0, red);
          }
          if (pass == 0) {
            wave_sum<3>(red);
            tot[0] += red[0]; tot[1] += red[1]; tot[2] += red[2];
          } else if (in) {
            const double4 q = core.backward(d, rec0, rec1, rec2, tot);
            if (hr > 0 && hr <= n_hot) {
              double* a = acc + (hr - 1);
              const double hx = rec0.x, hy = rec0.y, hz = rec0.z;
              const double v[12] = {hx * q.x, hy * q.x, hz * q.x, q.x, hx * q.y, hy * q.y,
                                    hz * q.y, q.y, hx * q.z, hy * q.z, hz * q.z, q.z};
#pragma unroll
              for (int k = 0; k < 12; ++k) __hip_atomic_fetch_add(a + k * n_hot, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              d.q4c[d.cold_pos[slot]] = q;
            }
          }
        }
      }
    }
  }
  if (ACC) {
    __syncthreads();
    // LDS holds acc[j][camera]; the partials go out as [camera][workgroup][12]: the per-camera sum of
    // cam_binv_axpy then reads one contiguous run of 96-byte records
    for (int i = threadIdx.x; i < n_hot * 12; i += E0C_BLOCK)
      hot_out[((size_t)(i / 12) * gridDim.x + blockIdx.x) * 12 + i % 12] = acc[(i % 12) * n_hot + i / 12];
  }
}

// ------------------------------------------------------------------------------------------
// K10, lane-per-landmark form (default of POVAR_E0_IMPLICIT_LDSACC, step 1): right_mul_e0_pOSE
// (linearization_power_varproj.hpp:364-406) with lane = landmark.  The per-landmark sum u = Jl^T t is a register
// accumulation over the rows of the tile (no wavefront scan, no segment metadata), v = Hll^-1 u is computed
// once per landmark instead of once per observation, and every global address is static, so the row stream is
// prefetched without dependent loads.  Per observation: 20 B from HBM (uv, camera rank), 72 B per landmark.
// Camera records (z_c, P_c[:, :3]) of the n_hot_acc most observed cameras live in LDS together with their
// Jp^T s accumulators, exactly as in e0_lm_cached<true>; observations of colder cameras gather the record from
// the rank-ordered image in L2 and leave their three scatter scalars in the cold camera-major view (q4c).
// ------------------------------------------------------------------------------------------
struct LplObs {
  double w, cu, cv, cuv;
  __device__ inline void set(const Dp& d, double2 uv, double w_) {
    const double sb2 = d.sb * d.sb;
    w = w_;
    cu = sb2 * uv.x;
    cv = sb2 * uv.y;
    cuv = sb2 * (uv.x * uv.x + uv.y * uv.y);
  }
};

// forward: red += P3^T (w C (Z h~))
__device__ inline void lpl_forward(const LplObs& o, const double* zz, const double* P3, double hx, double hy, double hz,
                                   double* red) {
  const double d0 = hx * zz[0] + hy * zz[1] + hz * zz[2] + zz[3];
  const double d1 = hx * zz[4] + hy * zz[5] + hz * zz[6] + zz[7];
  const double d2 = hx * zz[8] + hy * zz[9] + hz * zz[10] + zz[11];
  const double a0 = o.w * (d0 - o.cu * d2);
  const double a1 = o.w * (d1 - o.cv * d2);
  const double a2 = o.w * (o.cuv * d2 - o.cu * d0 - o.cv * d1);
  red[0] += P3[0] * a0 + P3[3] * a1 + P3[6] * a2;
  red[1] += P3[1] * a0 + P3[4] * a1 + P3[7] * a2;
  red[2] += P3[2] * a0 + P3[5] * a1 + P3[8] * a2;
}
// backward: q = w C (P3 g)
__device__ inline void lpl_backward(const LplObs& o, const double* P3, const double* g, double* q) {
  const double e0 = P3[0] * g[0] + P3[1] * g[1] + P3[2] * g[2];
  const double e1 = P3[3] * g[0] + P3[4] * g[1] + P3[5] * g[2];
  const double e2 = P3[6] * g[0] + P3[7] * g[1] + P3[8] * g[2];
  q[0] = o.w * (e0 - o.cu * e2);
  q[1] = o.w * (e1 - o.cv * e2);
  q[2] = o.w * (o.cuv * e2 - o.cu * e0 - o.cv * e1);
}
__device__ inline void lpl_read_zz(const double2* h, double* zz) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const double2 v = h[j];
    zz[2 * j] = v.x;
    zz[2 * j + 1] = v.y;
  }
}
__device__ inline void lpl_read_p3(const double2* h, double* P3) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double2 v = h[6 + j];
    P3[2 * j] = v.x;
    P3[2 * j + 1] = v.y;
  }
  P3[8] = h[10].x;
}

// The row stream of one wavefront is a static address sequence: tile t forward rows, tile t backward rows (the same
// rows again, now L2 hits), tile t + 16 ...  A scalar prefetch cursor runs LPL_DEPTH rows ahead of the consumer
// along that sequence, across the pass and tile boundaries, so the wavefront never waits for a load it has just
// issued (s_waitcnt vmcnt retires in issue order: the e0_lm_cached loop exposed two HBM latencies per bin that way).
// The next tile's landmark record is requested when the backward pass starts: G is dead by then (g = G u is
// formed), so only the three coordinates need a second set of registers.
constexpr int LPL_DEPTH = 3;
// Accumulator slots.  ds_add_f64 collisions inside a 32-lane half serialise, and the most observed cameras collect
// several observations per row (Zipf hub: 9 % of all observations): the LPL_HUBS hottest cameras therefore get four
// accumulator replicas each, chosen per observation by the layout (lpl_layout.hpp), summed at the flush.
#ifndef POVAR_LPL_HUBS
#define POVAR_LPL_HUBS 16
#endif
constexpr int LPL_HUBS = POVAR_LPL_HUBS;
__host__ __device__ inline int lpl_hubs(int n_hot) { return n_hot < LPL_HUBS ? n_hot : LPL_HUBS; }
// cw >= 0 packs the LDS slot (low 16 bits) and, for a hub, the accumulator replica the host chose (bits 16-17)
__host__ __device__ inline int lpl_cw_slot(int cw) { return cw & 0xffff; }
__host__ __device__ inline int lpl_acc_slot(int cw, int hubs) {
  const int slot = cw & 0xffff;
  return slot < hubs ? 4 * slot + ((cw >> 16) & 3) : slot + 3 * hubs;
}
__host__ __device__ inline size_t lpl_lds_bytes(int n_hot) {
  return (size_t)n_hot * HOT_REC * sizeof(double2) + (size_t)(n_hot + 3 * lpl_hubs(n_hot)) * 96 + 16;
}
// where a cold observation of row j, lane `lane` leaves its q in Dp::q4c (fl, nh: tile.w, tile.z; lpl_layout.hpp)
__device__ inline int lpl_cold_q(int fl, int nh, int j, int lane) { return ((fl >> 4) + (j - nh)) * WAVE + lane; }

struct LplRow {
  double2 uv;
  int cw;
  double w;
};
struct LplCursor {  // wave-uniform (SGPRs)
  int t, pass, j, row0, k;
};

template <bool ROBUST>
__global__ __launch_bounds__(E0C_BLOCK) void e0_lpl(Dp d, double* hot_out) {
  const int done = d.flags[1];  // requested first, tested after the LDS staging (no global side effects before)
  extern __shared__ double2 hot[];  // [n_hot][HOT_REC] records, then acc[12][n_slots], then the tile counter
  const V2& v = d.v2;
  // this workgroup's camera slots: the records of the cameras it keeps in LDS (lpl_layout.hpp) and their accumulators
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_hot = v.wg_cam_off[blockIdx.x + 1] - cam0;
  const int hubs = v.hubs, n_slots = n_hot + 3 * hubs;
  double* acc = reinterpret_cast<double*>(hot + n_hot * HOT_REC);
  int* grab_ctr = reinterpret_cast<int*>(acc + n_slots * 12);
  for (int i = threadIdx.x; i < n_slots * 12; i += E0C_BLOCK) acc[i] = 0;
  // the first tile of every wavefront is dealt statically (tiles are sorted longest first: the same tiles the first
  // sixteen grabs would return), the counter serves the later ones
  if (threadIdx.x == 0) *grab_ctr = E0C_BLOCK / WAVE;
  const double2* rec_img = reinterpret_cast<const double2*>(d.hot_rec);
  // staging, part 1: slot -> camera rank (the record pieces, a second dependent round trip, are requested further down:
  // the wavefront's first rows and landmark record go out in between, so the three latencies overlap instead of adding
  // up -- they are the fixed cost of a launch, a third of the kernel on an 8-GPU shard)
  constexpr int PASSES = (HOT_ACC_MAX * HOT_REC + E0C_BLOCK - 1) / E0C_BLOCK;
  int rk[PASSES];
#pragma unroll
  for (int u = 0; u < PASSES; ++u) {
    const int i = threadIdx.x + u * E0C_BLOCK;
    rk[u] = i < n_hot * HOT_REC ? v.wg_cams[cam0 + i / HOT_REC] : 0;
  }
  // where the accumulators go at the end (partial record of each slot): requested now, used after the last tile
  constexpr int FPASSES = (HOT_ACC_MAX * 6 + E0C_BLOCK - 1) / E0C_BLOCK;
  int frec[FPASSES];
#pragma unroll
  for (int u = 0; u < FPASSES; ++u) {
    const int i = threadIdx.x + u * E0C_BLOCK;
    frec[u] = i < n_hot * 6 ? v.wg_slot_rec[cam0 + i / 6] : 0;
  }
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
  {
    const long long t0 = (long long)t_begin + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    pc.t = t0 < t_end ? (int)t0 : t_end;
  }
  pc.pass = 0;
  pc.j = 0;
  pc.row0 = 0;
  pc.k = 1;
  int c_t = pc.t, c_row0 = 0, c_k = 0, c_nh = 0, c_fl = 0;
  // the tile after the one being consumed, taken when the consumer enters a tile (the first one after the staging
  // barrier below: the counter lives in LDS): the prefetch cursor runs at most LPL_DEPTH = 3 rows ahead and a tile has
  // at least 4 row steps, so it never needs more than this one -- and not before the consumer has started
  int nx_t = t_end;
  if (c_t < t_end) {
    tile_info(c_t, c_row0, c_k, c_nh, c_fl);
    pc.row0 = c_row0;
    pc.k = c_k;
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
  double hx = 0, hy = 0, hz = 0, G00 = 0, G01 = 0, G02 = 0, G11 = 0, G12 = 0, G22 = 0;
  if (c_t < t_end) {
    const double* rp = v.lmrec + ((size_t)c_t * 9) * WAVE + lane;
    hx = rp[0]; hy = rp[WAVE]; hz = rp[2 * WAVE];
    G00 = rp[3 * WAVE]; G01 = rp[4 * WAVE]; G02 = rp[5 * WAVE]; G11 = rp[6 * WAVE]; G12 = rp[7 * WAVE]; G22 = rp[8 * WAVE];
  }
  {
    // staging, part 2: the record pieces into LDS
    double2 piece[PASSES];
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int i = threadIdx.x + u * E0C_BLOCK;
      piece[u] = rec_img[(size_t)rk[u] * (HOT_REC_STRIDE / 2) + i % HOT_REC];
    }
#pragma unroll
    for (int u = 0; u < PASSES; ++u) {
      const int i = threadIdx.x + u * E0C_BLOCK;
      if (i < n_hot * HOT_REC) hot[i] = piece[u];
    }
  }
  __syncthreads();
  if (done) return;
  if (c_t < t_end) nx_t = grab();
  while (c_t < t_end) {
    double red[3] = {0, 0, 0};
    for (int j = 0; j < c_k; ++j) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      double zz[12], P3[9];
      LplObs o;
      o.set(d, cur.uv, ROBUST ? cur.w : 1.0);
      if (j < c_nh) {  // wave-uniform: every lane has an observation of an LDS-resident camera in this row
        const double2* h = hot + lpl_cw_slot(cur.cw) * HOT_REC;
        lpl_read_zz(h, zz);
        lpl_read_p3(h, P3);
        lpl_forward(o, zz, P3, hx, hy, hz, red);
      } else if (cur.cw != -1) {
        if (cur.cw >= 0) {
          const double2* h = hot + lpl_cw_slot(cur.cw) * HOT_REC;
          lpl_read_zz(h, zz);
          lpl_read_p3(h, P3);
        } else {  // cold: camera with popularity rank -2 - cw, record from the rank-ordered image (L2)
          const double2* h = rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2);
          lpl_read_zz(h, zz);
          lpl_read_p3(h, P3);
        }
        lpl_forward(o, zz, P3, hx, hy, hz, red);
      }
    }
    if (c_fl & 1) {  // landmarks dealt over several lanes: sum their partial u = Jl^T t (segmented wavefront scan)
      const int sg = v.seg[(size_t)c_t * WAVE + lane];
      seg_reduce_steps<3>(red, lane, sg & 255, (sg >> 8) & 255, 4);
    }
    const double g[3] = {G00 * red[0] + G01 * red[1] + G02 * red[2], G01 * red[0] + G11 * red[1] + G12 * red[2],
                         G02 * red[0] + G12 * red[1] + G22 * red[2]};
    // the next tile's record: G into its own (now dead) registers, the coordinates into a second set
    const int n_t = nx_t;
    double nhx = 0, nhy = 0, nhz = 0;
    if (n_t < t_end) {
      const double* rp = v.lmrec + ((size_t)n_t * 9) * WAVE + lane;
      nhx = rp[0]; nhy = rp[WAVE]; nhz = rp[2 * WAVE];
      G00 = rp[3 * WAVE]; G01 = rp[4 * WAVE]; G02 = rp[5 * WAVE]; G11 = rp[6 * WAVE]; G12 = rp[7 * WAVE]; G22 = rp[8 * WAVE];
    }
    for (int jj = 0; jj < c_k; ++jj) {
      const int j = c_k - 1 - jj;
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      double P3[9], q[3];
      LplObs o;
      o.set(d, cur.uv, ROBUST ? cur.w : 1.0);
      if (j < c_nh || cur.cw >= 0) {
        lpl_read_p3(hot + lpl_cw_slot(cur.cw) * HOT_REC, P3);
        lpl_backward(o, P3, g, q);
        double* a = acc + lpl_acc_slot(cur.cw, hubs);  // acc[m][slot]: consecutive slots on consecutive banks
        const double val[12] = {hx * q[0], hy * q[0], hz * q[0], q[0], hx * q[1], hy * q[1],
                                hz * q[1], q[1], hx * q[2], hy * q[2], hz * q[2], q[2]};
#pragma unroll
        for (int m = 0; m < 12; ++m)
          __hip_atomic_fetch_add(a + m * n_slots, val[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (cur.cw < -1) {
        // where q goes: row-major next to the other lanes' (graphs with many cold observations: the per-camera kernel
        // gathers, Dp::q_rows) or straight to its place in the camera-major cold view (few: one 32-byte store per lane)
        const int cold_at = d.q_rows ? lpl_cold_q(c_fl, c_nh, j, lane) : v.cpos[((size_t)c_row0 + j) * WAVE + lane];
        lpl_read_p3(rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2), P3);
        lpl_backward(o, P3, g, q);
        d.q4c[cold_at] = make_double4(q[0], q[1], q[2], 0);
      }
    }
    c_t = n_t;
    if (c_t < t_end) {
      tile_info(c_t, c_row0, c_k, c_nh, c_fl);
      nx_t = grab();
    }
    hx = nhx; hy = nhy; hz = nhz;
  }
  __syncthreads();
  // accumulators -> this workgroup's partial records (camera-major in hot_out: the per-camera kernel reads one run);
  // 16-byte stores; the record indices were requested in the prologue
  {
#pragma unroll
    for (int u = 0; u < FPASSES; ++u) {
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
        reinterpret_cast<double2*>(hot_out + (size_t)frec[u] * 12)[i % 6] = s;
      }
    }
  }
  if (d.p2p_epoch && blockIdx.x == 0 && threadIdx.x == 0) *d.p2p_epoch += 1;  // one tick per term, read by the next kernels
}

// ------------------------------------------------------------------------------------------
// K7 on the lane-per-landmark layout: get_Hll_inv_add_Hpp_b_pOSE / _poBA (landmark_block.hpp:510-572), the
// landmark half of prepare_Hb_pOSE.  Same structure as e0_lpl (lane = landmark, row stream with a three-row
// prefetch cursor, camera records and per-camera accumulators in LDS, cold observations through q4c): the forward
// pass accumulates Hll = Jl^T Jl and Jl^T r in registers, the lane then inverts Hll (+ lambda I for
// POWER_SCHUR_COMPLEMENT), stores Hll^-1 and the per-term landmark records, and the backward pass adds
// Jp^T (r - Jl w) into the camera accumulators.  The tile (Jl, r) is rebuilt from (P_c, x_l, u, v, sqrt(w), s_l):
// nothing per observation is read besides the 20-byte row.  Replaces lm_regular<OpPrepare> + cm_scatter +
// cam_sum_items (214 + 109 + 6 us on venice-1778) for the LDSACC mode.
// ------------------------------------------------------------------------------------------
constexpr int PREP_REC = 6;  // double2 per camera record: P[:, :3] row-major (9), then the translation column (3)
// LDS strides of the records are odd numbers of 16-byte quads: slot -> bank-quad class is then a bijection mod 16, the
// relation the row placement of the layout assumes (an even stride leaves 8, 4 or 2 classes: built-in conflicts)
constexpr int PREP_STRIDE = PREP_REC | 1;

struct PrepObs {
  double jl[12], r[4];
  // tile of one observation (bal_bundle_adjustment_helper.cpp:244-313 with the scalings of landmark_block.hpp:284-295)
  __device__ inline void set(const Dp& d, const double* P, double2 uv, double w, double hx, double hy, double hz,
                             double s0, double s1, double s2) {
    const double sw = sqrt(w), cb = d.sb * sw, ca = d.sa * sw;
    const double m0[3] = {P[0] - P[6] * uv.x, P[1] - P[7] * uv.x, P[2] - P[8] * uv.x};
    const double m1[3] = {P[3] - P[6] * uv.y, P[4] - P[7] * uv.y, P[5] - P[8] * uv.y};
    const double t0 = P[9] - P[11] * uv.x, t1 = P[10] - P[11] * uv.y;
    r[0] = cb * (m0[0] * hx + m0[1] * hy + m0[2] * hz + t0);
    r[1] = cb * (m1[0] * hx + m1[1] * hy + m1[2] * hz + t1);
    r[2] = ca * (P[0] * hx + P[1] * hy + P[2] * hz + P[9] - uv.x);
    r[3] = ca * (P[3] * hx + P[4] * hy + P[5] * hz + P[10] - uv.y);
    jl[0] = cb * m0[0] * s0; jl[1] = cb * m0[1] * s1; jl[2] = cb * m0[2] * s2;
    jl[3] = cb * m1[0] * s0; jl[4] = cb * m1[1] * s1; jl[5] = cb * m1[2] * s2;
    jl[6] = ca * P[0] * s0; jl[7] = ca * P[1] * s1; jl[8] = ca * P[2] * s2;
    jl[9] = ca * P[3] * s0; jl[10] = ca * P[4] * s1; jl[11] = ca * P[5] * s2;
  }
};
__device__ inline void prep_read_rec(const double2* h, double* P) {
#pragma unroll
  for (int j = 0; j < PREP_REC; ++j) {
    const double2 v = h[j];
    P[2 * j] = v.x;
    P[2 * j + 1] = v.y;
  }
}
__host__ __device__ inline size_t prep_lds_bytes(int n_hot) {
  return (size_t)n_hot * PREP_STRIDE * sizeof(double2) + (size_t)(n_hot + 3 * lpl_hubs(n_hot)) * 96 + 16;
}

template <bool ROBUST>
__global__ __launch_bounds__(E0C_BLOCK) void prepare_lpl(Dp d, double* hot_out) {
  extern __shared__ double2 hot[];  // [n_hot][PREP_REC] records, then acc[12][n_slots], then the tile counter
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
  LplRow n1, n2, n3;
  n1.cw = n2.cw = n3.cw = -1;
  n1.w = n2.w = n3.w = 1.0;
  n1.uv = n2.uv = n3.uv = make_double2(0, 0);
  issue(n1);
  issue(n2);
  issue(n3);
  while (c_t < t_end) {
    // the lane's landmark: coordinates and Jl column scale (gathers; once per tile)
    const int lm = v.lm_of[(size_t)c_t * WAVE + lane];
    const int sg = v.seg[(size_t)c_t * WAVE + lane];
    const double4 h4 = v.lml[(size_t)c_t * WAVE + lane], s4 = v.lsc[(size_t)c_t * WAVE + lane];
    const double hx = h4.x, hy = h4.y, hz = h4.z;
    double red[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < c_k; ++j) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      double P[12];
      if (cur.cw >= 0) prep_read_rec(hot + lpl_cw_slot(cur.cw) * PREP_STRIDE, P);
      else prep_read_rec(rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2) + 6, P);
      PrepObs o;
      o.set(d, P, cur.uv, ROBUST ? cur.w : 1.0, hx, hy, hz, s4.x, s4.y, s4.z);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[0] += o.jl[3 * r] * o.jl[3 * r];
        red[1] += o.jl[3 * r] * o.jl[3 * r + 1];
        red[2] += o.jl[3 * r] * o.jl[3 * r + 2];
        red[3] += o.jl[3 * r + 1] * o.jl[3 * r + 1];
        red[4] += o.jl[3 * r + 1] * o.jl[3 * r + 2];
        red[5] += o.jl[3 * r + 2] * o.jl[3 * r + 2];
        red[6] += o.jl[3 * r] * o.r[r];
        red[7] += o.jl[3 * r + 1] * o.r[r];
        red[8] += o.jl[3 * r + 2] * o.r[r];
      }
    }
    if (c_fl & 1) seg_reduce_steps<9>(red, lane, sg & 255, (sg >> 8) & 255, 4);
    double Hi[9], w3[3] = {0, 0, 0};
    if (lm >= 0) {
      double H[9];
      sym3(red, H);
      H[0] += d.lambda_lm;
      H[4] += d.lambda_lm;
      H[8] += d.lambda_lm;
      inv3(H, Hi);
      w3[0] = Hi[0] * red[6] + Hi[1] * red[7] + Hi[2] * red[8];
      w3[1] = Hi[3] * red[6] + Hi[4] * red[7] + Hi[5] * red[8];
      w3[2] = Hi[6] * red[6] + Hi[7] * red[7] + Hi[8] * red[8];
      // per-term record of this lane (e0_lpl); Hll^-1 and the row-major record of the other kernels once per landmark
      double* r2 = v.lmrec + ((size_t)c_t * 9) * WAVE + lane;
      r2[0] = hx; r2[WAVE] = hy; r2[2 * WAVE] = hz;
      r2[3 * WAVE] = s4.x * Hi[0] * s4.x; r2[4 * WAVE] = s4.x * Hi[1] * s4.y; r2[5 * WAVE] = s4.x * Hi[2] * s4.z;
      r2[6 * WAVE] = s4.y * Hi[4] * s4.y; r2[7 * WAVE] = s4.y * Hi[5] * s4.z; r2[8 * WAVE] = s4.z * Hi[8] * s4.z;
      if (lane == (sg & 255) && !d.prep_lpl_only) {
#pragma unroll
        for (int m = 0; m < 9; ++m) d.hll_inv[9 * (size_t)lm + m] = Hi[m];
        double4* rec = reinterpret_cast<double4*>(d.lmrec) + 3 * (size_t)lm;
        rec[0] = make_double4(hx, hy, hz, s4.x);
        rec[1] = make_double4(s4.y, s4.z, Hi[0], Hi[1]);
        rec[2] = make_double4(Hi[2], Hi[4], Hi[5], Hi[8]);
      }
    }
    for (int jj = 0; jj < c_k; ++jj) {
      const int j = c_k - 1 - jj;
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      double P[12];
      if (cur.cw >= 0) prep_read_rec(hot + lpl_cw_slot(cur.cw) * PREP_STRIDE, P);
      else prep_read_rec(rec_img + (size_t)(-2 - cur.cw) * (HOT_REC_STRIDE / 2) + 6, P);
      PrepObs o;
      const double w = ROBUST ? cur.w : 1.0;
      o.set(d, P, cur.uv, w, hx, hy, hz, s4.x, s4.y, s4.z);
      double e[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = o.r[r] - (o.jl[3 * r] * w3[0] + o.jl[3 * r + 1] * w3[1] + o.jl[3 * r + 2] * w3[2]);
      const double4 q = pose_q(d, cur.uv.x, cur.uv.y, sqrt(w), e);
      if (cur.cw >= 0) {
        double* a = acc + lpl_acc_slot(cur.cw, hubs);
        const double val[12] = {hx * q.x, hy * q.x, hz * q.x, q.x, hx * q.y, hy * q.y,
                                hz * q.y, q.y, hx * q.z, hy * q.z, hz * q.z, q.z};
#pragma unroll
        for (int m = 0; m < 12; ++m)
          __hip_atomic_fetch_add(a + m * n_slots, val[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        d.q4c[d.q_rows ? lpl_cold_q(c_fl, c_nh, j, lane) : v.cpos[((size_t)c_row0 + j) * WAVE + lane]] = make_double4(q.x, q.y, q.z, 0);
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

// ------------------------------------------------------------------------------------------
// K12 on the lane-per-landmark layout: back_substitute_pOSE (landmark_block.hpp:670-707), POWER_VARPROJ -- the
// arithmetic of OpBackVarproj on prepare_lpl's row stream.  Camera records in LDS: the UPDATED camera, its increment
// and the camera of the linearisation point (36 doubles); no accumulators.  First pass: H = Jl^T Jl and Jl^T res of
// the fresh, unweighted, unscaled tile; the lane solves for delta; second pass: the reference's model cost change
// (stored Jl and residual rebuilt from the linearisation point) summed per workgroup into part[blockIdx.x].
// Replaces lm_regular<OpBackVarproj> (204 us on venice-1778) for the LDSACC mode.
// ------------------------------------------------------------------------------------------
constexpr int BACK_REC = 18;  // double2 per camera record: P_new (12), inc (12), P_lin (12)
constexpr int BACK_STRIDE = BACK_REC | 1;
__host__ __device__ inline size_t back_lds_bytes(int n_hot) { return (size_t)n_hot * BACK_STRIDE * sizeof(double2) + 16; }

template <bool ROBUST>
__global__ __launch_bounds__(E0C_BLOCK) void backsub_lpl(Dp d, double* part) {
  extern __shared__ double2 hot[];  // [n_hot][BACK_REC] records, then the tile counter
  __shared__ double sh[E0C_BLOCK / 64];
  const V2& v = d.v2;
  const int cam0 = v.wg_cam_off[blockIdx.x];
  const int n_hot = v.wg_cam_off[blockIdx.x + 1] - cam0;
  int* grab_ctr = reinterpret_cast<int*>(hot + n_hot * BACK_STRIDE);
  if (threadIdx.x == 0) *grab_ctr = 0;
  // record piece j of a camera: 0-5 cams4, 6-11 inc, 12-17 cams_lin4 (each 12 doubles)
  auto piece = [&](int cam, int j) -> double2 {
    const double* src = j < 6 ? reinterpret_cast<const double*>(d.cams4) : j < 12 ? d.inc : reinterpret_cast<const double*>(d.cams_lin4);
    return reinterpret_cast<const double2*>(src + 12 * (size_t)cam)[j % 6];
  };
  for (int i = threadIdx.x; i < n_hot * BACK_REC; i += E0C_BLOCK) {
    const int r = i / BACK_REC, j = i - r * BACK_REC;
    hot[r * BACK_STRIDE + j] = piece(d.hot_cams[v.wg_cams[cam0 + r]], j);
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
  // 12 consecutive doubles of a record (LDS) or of a camera-indexed global array (cold camera)
  auto read12 = [&](const double2* hp, double4 (&o)[3]) {
    const double2 b0 = hp[0], b1 = hp[1], b2 = hp[2], b3 = hp[3], b4 = hp[4], b5 = hp[5];
    o[0] = make_double4(b0.x, b0.y, b1.x, b1.y);
    o[1] = make_double4(b2.x, b2.y, b3.x, b3.y);
    o[2] = make_double4(b4.x, b4.y, b5.x, b5.y);
  };
  // which: 0 P_new, 1 inc, 2 P_lin.  LDS pointer or global pointer, never a select of the two (a generic pointer turns
  // the reads into flat_loads)
  auto read_part = [&](int cw, int which, double4 (&o)[3]) {
    if (cw >= 0) {
      read12(hot + lpl_cw_slot(cw) * BACK_STRIDE + 6 * which, o);
    } else {
      const int cam = d.hot_cams[-2 - cw];
      const double* src = which == 0 ? reinterpret_cast<const double*>(d.cams4)
                                     : which == 1 ? d.inc : reinterpret_cast<const double*>(d.cams_lin4);
      read12(reinterpret_cast<const double2*>(src + 12 * (size_t)cam), o);
    }
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
    const size_t li = (size_t)c_t * WAVE + lane;
    const double4 h = v.lmx[li], hl = v.lml[li], s4 = v.lsc[li];
    double red[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < c_k; ++j) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      double4 pp[3];
      read_part(cur.cw, 0, pp);
      const Cam P = {pp[0], pp[1], pp[2]};
      double res[4], jl[12];
      pose_residual(d, P, h, cur.uv.x, cur.uv.y, res);
      pose_jl(d, P, cur.uv.x, cur.uv.y, 1.0, make_double4(1, 1, 1, 1), jl);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[0] += jl[3 * r] * jl[3 * r];
        red[1] += jl[3 * r] * jl[3 * r + 1];
        red[2] += jl[3 * r] * jl[3 * r + 2];
        red[3] += jl[3 * r + 1] * jl[3 * r + 1];
        red[4] += jl[3 * r + 1] * jl[3 * r + 2];
        red[5] += jl[3 * r + 2] * jl[3 * r + 2];
        red[6] += jl[3 * r] * res[r];
        red[7] += jl[3 * r + 1] * res[r];
        red[8] += jl[3 * r + 2] * res[r];
      }
    }
    if (c_fl & 1) seg_reduce_steps<9>(red, lane, sg & 255, (sg >> 8) & 255, 4);
    double dl[3] = {0, 0, 0};
    if (lm >= 0) {
      double H[9], Hi[9];
      sym3(red, H);
      inv3(H, Hi);
      dl[0] = -(Hi[0] * red[6] + Hi[1] * red[7] + Hi[2] * red[8]);
      dl[1] = -(Hi[3] * red[6] + Hi[4] * red[7] + Hi[5] * red[8]);
      dl[2] = -(Hi[6] * red[6] + Hi[7] * red[7] + Hi[8] * red[8]);
      const double4 hn = make_double4(h.x + dl[0], h.y + dl[1], h.z + dl[2], h.w);
      v.lmx[li] = hn;  // the mirror stays current: no rebuild after an accepted step
      if (lane == (sg & 255)) d.lms4[lm] = hn;
    }
    for (int jj = 0; jj < c_k; ++jj) {
      const LplRow cur = n1;
      n1 = n2;
      n2 = n3;
      issue(n3);
      if (cur.cw == -1) continue;
      double4 zz[3], pl[3];
      read_part(cur.cw, 1, zz);
      read_part(cur.cw, 2, pl);
      const Cam Pl = {pl[0], pl[1], pl[2]};
      const double sw = ROBUST ? sqrt(cur.w) : 1.0;
      double jinc[4], jls[12], rr[4];
      pose_jp_x(d, h, cur.uv.x, cur.uv.y, 1.0, zz, jinc);
      pose_jl(d, Pl, cur.uv.x, cur.uv.y, sw, s4, jls);
      pose_residual(d, Pl, hl, cur.uv.x, cur.uv.y, rr);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double ji = jinc[r] + (jls[3 * r] * dl[0] + jls[3 * r + 1] * dl[1] + jls[3 * r + 2] * dl[2]);
        sc -= ji * (0.5 * ji + sw * rr[r]);
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

// ------------------------------------------------------------------------------------------
// K2 / K3 + K5 on the lane-per-landmark layout: one forward walk over the rows, camera matrices in LDS (96 B per slot),
// no accumulators.
//   MODE 0  linearize_landmark_pOSE + scale_Jl_cols_pOSE (landmark_block.hpp:135-178, 284-295) at (cams_lin4, lms_lin4):
//           robust weight per observation (V2::w), Jl column scale per landmark, finiteness flag.  The per-slot
//           sqrt(w) / weighted residual arrays of the lane-per-observation kernels are NOT written: the lane-per-landmark
//           kernels rebuild both from (P, x, u, v, w); povar_lm.hip fills them when a legacy kernel asks (ensure_legacy).
//   MODE 1  compute_error_pOSE (bal_bundle_adjustment_helper.cpp:117-154) at (cams4, lms4): (error, |r|, count) summed per
//           workgroup into part[3 * blockIdx.x ..].
// ------------------------------------------------------------------------------------------
constexpr int PASS_REC = 6;  // double2 per camera record: P row-major
constexpr int PASS_STRIDE = PASS_REC | 1;
__host__ __device__ inline size_t pass_lds_bytes(int n_hot) { return (size_t)n_hot * PASS_STRIDE * sizeof(double2) + 16; }

template <int MODE>
__global__ __launch_bounds__(E0C_BLOCK) void lpl_pass(Dp d, double* part) {
  extern __shared__ double2 hot[];  // [n_hot][PASS_REC] records, then the tile counter
  __shared__ double sh[3 * (E0C_BLOCK / 64)];
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
  // single pass: the prefetch cursor walks tile after tile, three rows ahead (a tile has at least two rows here, so the
  // cursor may need the tile after the next one: two tiles are taken ahead)
  int c_t = grab(), q1 = c_t < t_end ? grab() : t_end, q2 = q1 < t_end ? grab() : t_end;
  int pc_t = c_t, pc_ahead = 0, pc_j = 0, pc_row0 = 0, pc_k = 1;  // pc_ahead: 0 = c_t, 1 = q1, 2 = q2
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
  double sc[3] = {0, 0, 0};
  int bad = 0;
  while (c_t < t_end) {
    const int c_row0 = tiles[4 * c_t], c_k = tiles[4 * c_t + 1], c_fl = tiles[4 * c_t + 3];
    const double4 h = v.lmx[(size_t)c_t * WAVE + lane];
    if (MODE == 0) v.lml[(size_t)c_t * WAVE + lane] = h;  // the linearisation point, lane-ordered, is left behind
    double red[3] = {0, 0, 0};
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
      double res[4];
      pose_residual(d, P, h, cur.uv.x, cur.uv.y, res);
      const double r2 = res[0] * res[0] + res[1] * res[1] + res[2] * res[2] + res[3] * res[3];
      double e, w;
      error_weight(d, r2, e, w);
      if (MODE == 0) {
        const double sw = sqrt(w);
        bad |= !isfinite(r2) || !isfinite(sw);
        if (d.robust && v.w) v.w[((size_t)c_row0 + j) * WAVE + lane] = w;
        double jl[12];
        pose_jl(d, P, cur.uv.x, cur.uv.y, sw, make_double4(1, 1, 1, 1), jl);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[0] += jl[3 * r] * jl[3 * r];
          red[1] += jl[3 * r + 1] * jl[3 * r + 1];
          red[2] += jl[3 * r + 2] * jl[3 * r + 2];
        }
      } else {
        bad |= !isfinite(r2);
        sc[0] += e;
        sc[1] += sqrt(r2);
        sc[2] += 1.0;
      }
    }
    if (MODE == 0) {
      const int sg = v.seg[(size_t)c_t * WAVE + lane];
      if (c_fl & 1) seg_reduce_steps<3>(red, lane, sg & 255, (sg >> 8) & 255, 4);
      const double4 sc4 = d.scale_jl ? make_double4(1.0 / (d.eps + sqrt(red[0])), 1.0 / (d.eps + sqrt(red[1])),
                                                    1.0 / (d.eps + sqrt(red[2])), 0.0)
                                     : make_double4(1.0, 1.0, 1.0, 0.0);
      // every lane of the landmark holds the segment total.  The landmark-order copy (Dp::jl_scale4) is not written
      // here: lanes_to_lm fills it when a lane-per-observation kernel or an export asks (povar_lm.hip: ensure_legacy)
      v.lsc[(size_t)c_t * WAVE + lane] = sc4;
    }
    c_t = q1;
    q1 = q2;
    q2 = q1 < t_end ? grab() : t_end;
    --pc_ahead;
  }
  if (bad) atomicOr(&d.flags[0], 1);
  if (MODE == 1) {
    block_sum<3, E0C_BLOCK>(sc, sh);
    if (threadIdx.x == 0) {
      part[3 * (size_t)blockIdx.x] = sc[0];
      part[3 * (size_t)blockIdx.x + 1] = sc[1];
      part[3 * (size_t)blockIdx.x + 2] = sc[2];
    }
  }
}

// K10 (stored tiles): right_mul_e0_pOSE on the tiles kept in HBM, blocked layout
// tiles[bin][pair][lane] (double2): pairs 0-23 Jp (row-major 4x12), 24-29 Jl (4x3), 30-31 r.
// Every byte of a tile is read once per term, 16 B per lane, 1 KiB contiguous per wave
// instruction.  The forward products (Jp x, Jl^T t, Jl v) use the stored values; the scatter
// Jp^T s (the reference's mutex-guarded +=, linearization_power_varproj.hpp:393-397) goes through
// the same three scalars per observation + camera-major pass as the implicit variant: a first
// version with one hardware fp64 atomic per output ran 45 ms per term on venice-1778 (hub cameras
// with 4e5 observations serialise), profiles/r01_a_first_kernel_stats.csv.
struct OpE0Tiles {
  static constexpr int NRED = 3, NSC = 0;
  static constexpr bool CHECK_DONE = true;
  struct Local {
    double jp[48];
    double jl[12];
  };
  __device__ void phase1(const Dp& d, int slot, int cam, int, double2, Local& L, double* red) const {
    const double2* t = d.tiles + ((size_t)(slot >> 6) * TILE_PAIRS) * WAVE + (slot & 63);
#pragma unroll
    for (int p = 0; p < 24; ++p) {
      const double2 a = t[p * WAVE];
      L.jp[2 * p] = a.x;
      L.jp[2 * p + 1] = a.y;
    }
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const double2 a = t[(24 + p) * WAVE];
      L.jl[2 * p] = a.x;
      L.jl[2 * p + 1] = a.y;
    }
    const double* x = d.tmp + 12 * (size_t)cam;
    double xc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) xc[j] = x[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double tr = 0;
#pragma unroll
      for (int j = 0; j < 12; ++j) tr += L.jp[12 * r + j] * xc[j];
      red[0] += L.jl[3 * r] * tr;
      red[1] += L.jl[3 * r + 1] * tr;
      red[2] += L.jl[3 * r + 2] * tr;
    }
  }
  __device__ void phase2(const Dp& d, int slot, int, int lm, double2 uv, Local& L, const double* tot,
                         double*) const {
    const double* Hi = d.hll_inv + 9 * (size_t)lm;
    const double v0 = Hi[0] * tot[0] + Hi[1] * tot[1] + Hi[2] * tot[2];
    const double v1 = Hi[3] * tot[0] + Hi[4] * tot[1] + Hi[5] * tot[2];
    const double v2 = Hi[6] * tot[0] + Hi[7] * tot[1] + Hi[8] * tot[2];
    double s[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = L.jl[3 * r] * v0 + L.jl[3 * r + 1] * v1 + L.jl[3 * r + 2] * v2;
    const double sw = d.robust ? d.sw[slot] : 1.0;
    store_q(d, slot, pose_q(d, uv.x, uv.y, sw, s));
  }
  __device__ void finish_lm(const Dp&, int, const double*) const {}
};

// Stored-tile E0 with the LDS treatment of e0_lm_cached<true> (POVAR_E0_TILES_LDSACC): one 768-thread
// workgroup per CU streams its bins' tiles (480 B per observation, every byte once); x_c of the hot
// cameras comes from LDS and their Jp^T s -- computed from the STORED Jp, all 48 values -- is
// accumulated in LDS (ds_add_f64) instead of going through q4 + cm_scatter.  Colder cameras and
// >64-observation landmarks keep the q4 path of OpE0Tiles.
constexpr int E0T_BLOCK = 768;
constexpr int HOT_REC_T = 6;  // double2 per cached camera: x_c (12 doubles)
POVAR_KERNEL __launch_bounds__(E0T_BLOCK) void e0_tiles_cached(Dp d, int bins_per_wg, double* hot_out) {
  if (d.flags[1]) return;
  extern __shared__ double2 hot[];  // [n_hot][HOT_REC_T] then acc[12][n_hot]
  const int n_hot = d.n_hot_acc;
  double* acc = reinterpret_cast<double*>(hot + n_hot * HOT_REC_T);
  for (int i = threadIdx.x; i < n_hot * 12; i += E0T_BLOCK) acc[i] = 0;
  {
    const double2* x2 = reinterpret_cast<const double2*>(d.tmp);
    for (int i = threadIdx.x; i < n_hot * HOT_REC_T; i += E0T_BLOCK) {
      const int r = i / HOT_REC_T, j = i - r * HOT_REC_T;
      hot[i] = x2[6 * (size_t)d.hot_cams[r] + j];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bin0 = blockIdx.x * bins_per_wg;
  const int bin1 = min(bin0 + bins_per_wg, d.n_bins);
  for (int bin = bin0 + wave; bin < bin1; bin += E0T_BLOCK / WAVE) {
    const int slot = bin * WAVE + lane;
    const int meta = d.meta[slot];
    const bool valid = (meta & META_REAL) && !(meta & META_LONG);
    const int seg_first = meta & 255, seg_last = (meta >> 8) & 255;
    const int hr = ((meta >> META_HOT_SHIFT) & META_HOT_MASK);
    const bool is_hot = hr > 0 && hr <= n_hot;
    double jp[48], jl[12];
    double red[3] = {0, 0, 0};
    int cam = 0, lm = 0;
    if (valid) {
      cam = d.cam[slot];
      lm = d.lm[slot];
      const double2* t = d.tiles + ((size_t)bin * TILE_PAIRS) * WAVE + lane;
#pragma unroll
      for (int p = 0; p < 24; ++p) {
        const double2 a = t[p * WAVE];
        jp[2 * p] = a.x;
        jp[2 * p + 1] = a.y;
      }
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const double2 a = t[(24 + p) * WAVE];
        jl[2 * p] = a.x;
        jl[2 * p + 1] = a.y;
      }
      double xc[12];
      if (is_hot) {
        const double2* h = hot + (hr - 1) * HOT_REC_T;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const double2 v = h[j];
          xc[2 * j] = v.x;
          xc[2 * j + 1] = v.y;
        }
      } else {
        const double2* x2 = reinterpret_cast<const double2*>(d.tmp) + 6 * (size_t)cam;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const double2 v = x2[j];
          xc[2 * j] = v.x;
          xc[2 * j + 1] = v.y;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double tr = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) tr += jp[12 * r + j] * xc[j];
        red[0] += jl[3 * r] * tr;
        red[1] += jl[3 * r + 1] * tr;
        red[2] += jl[3 * r + 2] * tr;
      }
    }
    seg_reduce_steps<3>(red, lane, seg_first, seg_last,
                        __builtin_amdgcn_readfirstlane((meta >> META_STEPS_SHIFT) & 7));
    if (valid) {
      const double* Hi = d.hll_inv + 9 * (size_t)lm;
      const double v0 = Hi[0] * red[0] + Hi[1] * red[1] + Hi[2] * red[2];
      const double v1 = Hi[3] * red[0] + Hi[4] * red[1] + Hi[5] * red[2];
      const double v2 = Hi[6] * red[0] + Hi[7] * red[1] + Hi[8] * red[2];
      double s[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] = jl[3 * r] * v0 + jl[3 * r + 1] * v1 + jl[3 * r + 2] * v2;
      if (is_hot) {
        double* a = acc + (hr - 1);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const double o = jp[j] * s[0] + jp[12 + j] * s[1] + jp[24 + j] * s[2] + jp[36 + j] * s[3];
          __hip_atomic_fetch_add(a + j * n_hot, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        const double2 uv = d.uv[slot];
        store_q(d, slot, pose_q(d, uv.x, uv.y, d.robust ? d.sw[slot] : 1.0, s));
      }
    }
  }
  __syncthreads();
  // the stored Jp carries the pose scaling sigma; the partials leave in the unscaled convention of
  // the scatter items (cam_binv_axpy multiplies by sigma once for both)
  for (int i = threadIdx.x; i < n_hot * 12; i += E0T_BLOCK) {
    const int cam_r = i / 12, j = i % 12;
    hot_out[((size_t)cam_r * gridDim.x + blockIdx.x) * 12 + j] =
        acc[j * n_hot + cam_r] / d.sigma[12 * (size_t)d.hot_cams[cam_r] + j];
  }
}

// K12: back_substitute_pOSE (landmark_block.hpp:670-707), POWER_VARPROJ.  Fresh unweighted,
// unscaled res/Jl at the UPDATED cameras, exact landmark re-solve, and the reference's model
// cost change with its mixture of scaled and unscaled quantities.
struct OpBackVarproj {
  static constexpr int NRED = 9, NSC = 1;
  static constexpr bool CHECK_DONE = false;
  struct Local {
    double jinc[4];  // Jp_fresh * inc
    double jls[12];  // stored Jl (weighted, column-scaled, old cameras)
    double4 r;
  };
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local& L, double* red) const {
    const Cam P = load_cam(d.cams4, cam);
    const double4 h = d.lms4[lm];
    double res[4];
    pose_residual(d, P, h, uv.x, uv.y, res);
    double jl[12];
    pose_jl(d, P, uv.x, uv.y, 1.0, make_double4(1, 1, 1, 1), jl);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[0] += jl[3 * r] * jl[3 * r];
      red[1] += jl[3 * r] * jl[3 * r + 1];
      red[2] += jl[3 * r] * jl[3 * r + 2];
      red[3] += jl[3 * r + 1] * jl[3 * r + 1];
      red[4] += jl[3 * r + 1] * jl[3 * r + 2];
      red[5] += jl[3 * r + 2] * jl[3 * r + 2];
      red[6] += jl[3 * r] * res[r];
      red[7] += jl[3 * r + 1] * res[r];
      red[8] += jl[3 * r + 2] * res[r];
    }
    const double4* ic = reinterpret_cast<const double4*>(d.inc) + 3 * cam;
    const double4 zz[3] = {ic[0], ic[1], ic[2]};
    pose_jp_x(d, h, uv.x, uv.y, 1.0, zz, L.jinc);
    const Cam Pl = load_cam(d.cams_lin4, cam);
    const double sw = d.robust ? d.sw[slot] : 1.0;
    pose_jl(d, Pl, uv.x, uv.y, sw, d.jl_scale4[lm], L.jls);
    L.r = d.rres[slot];
  }
  __device__ static void delta(const double* tot, double (&dl)[3]) {
    double H[9], Hi[9];
    sym3(tot, H);
    inv3(H, Hi);
    dl[0] = -(Hi[0] * tot[6] + Hi[1] * tot[7] + Hi[2] * tot[8]);
    dl[1] = -(Hi[3] * tot[6] + Hi[4] * tot[7] + Hi[5] * tot[8]);
    dl[2] = -(Hi[6] * tot[6] + Hi[7] * tot[7] + Hi[8] * tot[8]);
  }
  __device__ void phase2(const Dp&, int, int, int, double2, Local& L, const double* tot, double* sc) const {
    double dl[3];
    delta(tot, dl);
    const double rr[4] = {L.r.x, L.r.y, L.r.z, L.r.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double ji = L.jinc[r] + (L.jls[3 * r] * dl[0] + L.jls[3 * r + 1] * dl[1] + L.jls[3 * r + 2] * dl[2]);
      sc[0] -= ji * (0.5 * ji + rr[r]);
    }
  }
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    double dl[3];
    delta(tot, dl);
    double4 x = d.lms4[lm];
    x.x += dl[0];
    x.y += dl[1];
    x.z += dl[2];
    d.lms4[lm] = x;
  }
};

// K12: back_substitute_poBA (landmark_block.hpp:625-656), POWER_SCHUR_COMPLEMENT: stored tiles,
// scaled increment (d.z = sigma * inc), damped Hll, delta scaled by Jl_col_scale on update.
struct OpBackPoba {
  static constexpr int NRED = 9, NSC = 1;
  static constexpr bool CHECK_DONE = false;
  struct Local {
    double jpi[4];
    double jl[12];
    double4 r;
  };
  __device__ void phase1(const Dp& d, int slot, int cam, int lm, double2 uv, Local& L, double* red) const {
    const Cam P = load_cam(d.cams_lin4, cam);
    const double4 h = d.lms_lin4[lm];
    const double sw = d.robust ? d.sw[slot] : 1.0;
    pose_jl(d, P, uv.x, uv.y, sw, d.jl_scale4[lm], L.jl);
    const double4* zc = reinterpret_cast<const double4*>(d.z) + 3 * cam;
    const double4 zz[3] = {zc[0], zc[1], zc[2]};
    pose_jp_x(d, h, uv.x, uv.y, sw, zz, L.jpi);
    L.r = d.rres[slot];
    const double rr[4] = {L.r.x, L.r.y, L.r.z, L.r.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[0] += L.jl[3 * r] * L.jl[3 * r];
      red[1] += L.jl[3 * r] * L.jl[3 * r + 1];
      red[2] += L.jl[3 * r] * L.jl[3 * r + 2];
      red[3] += L.jl[3 * r + 1] * L.jl[3 * r + 1];
      red[4] += L.jl[3 * r + 1] * L.jl[3 * r + 2];
      red[5] += L.jl[3 * r + 2] * L.jl[3 * r + 2];
      const double a = rr[r] + L.jpi[r];
      red[6] += L.jl[3 * r] * a;
      red[7] += L.jl[3 * r + 1] * a;
      red[8] += L.jl[3 * r + 2] * a;
    }
  }
  __device__ static void delta(const Dp& d, const double* tot, double (&dl)[3]) {
    double H[9], Hi[9];
    sym3(tot, H);
    H[0] += d.lambda_lm;
    H[4] += d.lambda_lm;
    H[8] += d.lambda_lm;
    inv3(H, Hi);
    dl[0] = -(Hi[0] * tot[6] + Hi[1] * tot[7] + Hi[2] * tot[8]);
    dl[1] = -(Hi[3] * tot[6] + Hi[4] * tot[7] + Hi[5] * tot[8]);
    dl[2] = -(Hi[6] * tot[6] + Hi[7] * tot[7] + Hi[8] * tot[8]);
  }
  __device__ void phase2(const Dp& d, int, int, int, double2, Local& L, const double* tot, double* sc) const {
    double dl[3];
    delta(d, tot, dl);
    const double rr[4] = {L.r.x, L.r.y, L.r.z, L.r.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double ji = L.jpi[r] + (L.jl[3 * r] * dl[0] + L.jl[3 * r + 1] * dl[1] + L.jl[3 * r + 2] * dl[2]);
      sc[0] -= ji * (0.5 * ji + rr[r]);
    }
  }
  __device__ void finish_lm(const Dp& d, int lm, const double* tot) const {
    double dl[3];
    delta(d, tot, dl);
    const double4 s = d.jl_scale4[lm];
    double4 x = d.lms4[lm];
    x.x += dl[0] * s.x;
    x.y += dl[1] * s.y;
    x.z += dl[2] * s.z;
    d.lms4[lm] = x;
  }
};

// K3/K5/K6 materialised: the reference's stored tile (after Jl and Jp column scaling) of every
// observation in the blocked layout used by OpE0Tiles and exported by povar_get_buffer.
POVAR_KERNEL __launch_bounds__(LM_BLOCK) void materialize_tiles(Dp d) {
  const int slot = blockIdx.x * LM_BLOCK + threadIdx.x;
  if (slot >= d.n_bins * WAVE) return;
  double2* t = d.tiles + ((size_t)(slot >> 6) * TILE_PAIRS) * WAVE + (slot & 63);
  const int meta = d.meta[slot];
  if (!(meta & META_REAL)) {
#pragma unroll
    for (int p = 0; p < TILE_PAIRS; ++p) t[p * WAVE] = make_double2(0, 0);
    return;
  }
  const int cam = d.cam[slot], lm = d.lm[slot];
  const double2 uv = d.uv[slot];
  const Cam P = load_cam(d.cams_lin4, cam);
  const double4 h = d.lms_lin4[lm];
  const double sw = d.robust ? d.sw[slot] : 1.0;
  const double hh[4] = {h.x, h.y, h.z, h.w};
  const double* sg = d.sigma + 12 * (size_t)cam;
  double jp[48];
#pragma unroll
  for (int j = 0; j < 48; ++j) jp[j] = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // ((h * sb) * sw) * sigma: the order the reference applies them (helper.cpp:273-303,
    // landmark_block.hpp:167, 330-332)
    jp[j] = hh[j] * d.sb * sw * sg[j];
    jp[8 + j] = -hh[j] * uv.x * d.sb * sw * sg[8 + j];
    jp[12 + 4 + j] = hh[j] * d.sb * sw * sg[4 + j];
    jp[12 + 8 + j] = -hh[j] * uv.y * d.sb * sw * sg[8 + j];
    jp[24 + j] = hh[j] * d.sa * sw * sg[j];
    jp[36 + 4 + j] = hh[j] * d.sa * sw * sg[4 + j];
  }
  double jl[12];
  pose_jl(d, P, uv.x, uv.y, sw, d.jl_scale4[lm], jl);
  const double4 r = d.rres[slot];
#pragma unroll
  for (int p = 0; p < 24; ++p) t[p * WAVE] = make_double2(jp[2 * p], jp[2 * p + 1]);
#pragma unroll
  for (int p = 0; p < 6; ++p) t[(24 + p) * WAVE] = make_double2(jl[2 * p], jl[2 * p + 1]);
  t[30 * WAVE] = make_double2(r.x, r.y);
  t[31 * WAVE] = make_double2(r.z, r.w);
}

// ------------------------------------------------------------------------------------------
// camera-major kernels
// ------------------------------------------------------------------------------------------

// Second half of every Jp^T(.) product: item_part[item] = sum over the item's observations of
// ( h q0 ; h q1 ; h q2 ).  One wavefront per item, fixed order.
POVAR_KERNEL __launch_bounds__(256) void cm_scatter(Dp d, int check_done, int hom) {
  if (check_done && d.flags[1]) return;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (item >= d.cmv.n_items) return;
  const int b = d.cmv.item_off[item], e = d.cmv.item_off[item + 1];
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0;
  // 4 observations per lane in flight: index loads, then the dependent 32-byte gathers, then the FMAs
  constexpr int U = 4;
  for (int p0 = b + lane; p0 < e; p0 += U * WAVE) {
    int sl[U];
    double hx[U], hy[U], hz[U], hw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u * WAVE;
      const bool in = p < e;
      const int pc = in ? p : b;
      sl[u] = in ? d.cmv.slot[pc] : -1;
      hx[u] = d.cmv.h[pc];
      hy[u] = d.cmv.h[d.cmv.n + pc];
      hz[u] = d.cmv.h[2 * d.cmv.n + pc];
      hw[u] = hom ? d.cmv.h[3 * d.cmv.n + pc] : 1.0;  // step 2: homogeneous landmark (X0..X3)
    }
    double4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = sl[u] >= 0 ? d.q4[sl[u]] : make_double4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] += hx[u] * q[u].x; acc[1] += hy[u] * q[u].x; acc[2] += hz[u] * q[u].x; acc[3] += hw[u] * q[u].x;
      acc[4] += hx[u] * q[u].y; acc[5] += hy[u] * q[u].y; acc[6] += hz[u] * q[u].y; acc[7] += hw[u] * q[u].y;
      acc[8] += hx[u] * q[u].z; acc[9] += hy[u] * q[u].z; acc[10] += hz[u] * q[u].z; acc[11] += hw[u] * q[u].z;
    }
  }
  wave_sum<12>(acc);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 12; ++k) d.cmv.part[12 * (size_t)item + k] = acc[k];
  }
}

// landmark coordinates at the linearisation point, copied into camera-major order once per
// linearisation so the per-term camera-major pass streams them instead of gathering
POVAR_KERNEL __launch_bounds__(256) void cm_build_h(Dp d, const int* lm_of, double* out, int64_t n, int hom) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const double4 h = d.lms_lin4[lm_of[p]];
  out[p] = h.x;
  out[n + p] = h.y;
  out[2 * n + p] = h.z;
  if (hom) out[3 * n + p] = h.w;
}

// Camera-block Gram sums of the unscaled weighted Jp: Jp^T Jp = w * (C (x) h h^T) with
// C = [[1,0,-sb^2 u],[0,1,-sb^2 v],[.,.,sb^2(u^2+v^2)]] (sa^2 + sb^2 = 1), so four weighted
// moments of h h^T (10 unique entries each) per camera carry both get_Jp_diag2_pOSE
// (linearization_varproj.hpp:183-222) and the Hpp blocks (landmark_block.hpp:530-536).
POVAR_KERNEL __launch_bounds__(256) void cm_gram(Dp d, int gather) {
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (item >= d.n_items) return;
  const int b = d.item_off[item], e = d.item_off[item + 1];
  double acc[40];
#pragma unroll
  for (int k = 0; k < 40; ++k) acc[k] = 0;
  const Cam Pc = load_cam(d.cams_lin4, d.item_cam[item]);
  for (int p = b + lane; p < e; p += WAVE) {
    // gather: the camera-major copy of the landmarks and the per-slot sqrt(w) are not kept (lane-per-landmark mode):
    // the landmark is read in place and the robust weight recomputed from the residual
    double4 h;
    if (gather) {
      h = d.lms_lin4[d.cm_lm[p]];
      h.w = 1.0;
    } else {
      h = make_double4(d.cm_h[p], d.cm_h[d.n_obs + p], d.cm_h[2 * d.n_obs + p], 1.0);
    }
    const double2 uv = d.cm_uv[p];
    double w = 1.0;
    if (d.robust && gather) {
      double res[4], e_;
      pose_residual(d, Pc, h, uv.x, uv.y, res);
      error_weight(d, res[0] * res[0] + res[1] * res[1] + res[2] * res[2] + res[3] * res[3], e_, w);
      const double sw = sqrt(w);  // the reference squares the stored sqrt(w) (landmark_block.hpp:162-169)
      w = sw * sw;
    } else if (d.robust) {
      const double sw = d.sw[d.cm_slot[p]];
      w = sw * sw;
    }
    const double m[4] = {w, w * uv.x, w * uv.y, w * (uv.x * uv.x + uv.y * uv.y)};
    const double hh[10] = {h.x * h.x, h.x * h.y, h.x * h.z, h.x * h.w, h.y * h.y,
                           h.y * h.z, h.y * h.w, h.z * h.z, h.z * h.w, h.w * h.w};
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

// ------------------------------------------------------------------------------------------
// camera kernels
// ------------------------------------------------------------------------------------------

__device__ inline int sym10(int i, int j) {  // index of (i,j) in the packed upper 4x4
  if (i > j) { const int t = i; i = j; j = t; }
  return i * 4 - (i * (i - 1)) / 2 + (j - i);
}

// per camera: G = sum of item Gram parts; diag2 and pose scaling (linearizor_power_varproj.cpp:62-70)
constexpr int CFL_THREADS = 1024;  // sixteen item streams per camera: the hub camera (880 items on venice) is the tail of this launch (256 threads: 56 instead of 18 us)
POVAR_KERNEL __launch_bounds__(CFL_THREADS) void cam_finish_linearize(Dp d, const double* G_in) {
  const int c = blockIdx.x;
  constexpr int NQ = CFL_THREADS / 64;
  __shared__ double part[NQ][40];
  __shared__ double g[40];
  if (G_in) {
    if (threadIdx.x < 40) g[threadIdx.x] = G_in[40 * (size_t)c + threadIdx.x];
  } else {
    // thread (q, e): every 16th item of the camera starting at q, element e; fixed order
    const int e = threadIdx.x % 64, q = threadIdx.x / 64;
    if (e < 40) {
      double s = 0;
      for (int it = d.cam_item_off[c] + q; it < d.cam_item_off[c + 1]; it += NQ)
        s += d.item_partG[40 * (size_t)it + e];
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
    const double v = blk < 2 ? g[dj] : d.sb * d.sb * g[30 + dj];
    d.diag2[12 * (size_t)c + threadIdx.x] = v;
    d.sigma[12 * (size_t)c + threadIdx.x] = 1.0 / (d.eps + sqrt(v));
  }
}

// Cholesky inverse of a small SPD matrix by the 16 lanes that own it (LDS, row stride 12; reads the upper triangle,
// leaves L in the lower one): column j of L in two lock steps (the diagonal by lane j, the rest by lanes i > j), then
// lane `col` solves L L^T x = e_col in registers.  Same operation order as the serial form
// (linearization_power_varproj.hpp:145-148: llt().solve(Identity)) -- bit-identical results, 12 x the lanes.
// Every thread of the workgroup must call it (barriers); out: N x N row-major or nullptr.
template <int N>
__device__ inline void chol_inverse_16(double* A, int l, double* out) {
  for (int j = 0; j < N; ++j) {
    if (l == j) {
      double dd = A[j * 12 + j];
      for (int k = 0; k < j; ++k) dd -= A[j * 12 + k] * A[j * 12 + k];
      A[j * 12 + j] = sqrt(dd);
    }
    __syncthreads();
    if (l > j && l < N) {
      double sv = A[j * 12 + l];
      for (int k = 0; k < j; ++k) sv -= A[l * 12 + k] * A[j * 12 + k];
      A[l * 12 + j] = sv / A[j * 12 + j];
    }
    __syncthreads();
  }
  if (l < N) {
    double x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double sv = (i == l) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) sv -= A[i * 12 + k] * x[k];
      x[i] = sv / A[i * 12 + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
      double sv = x[i];
#pragma unroll
      for (int k = i + 1; k < N; ++k) sv -= A[k * 12 + i] * x[k];
      x[i] = sv / A[i * 12 + i];
    }
    if (out) {
#pragma unroll
      for (int i = 0; i < N; ++i) out[N * i + l] = x[i];
    }
  }
}

// K8: B_c = Hpp_c + lambda I, B_c^-1 by Cholesky (upper triangle) and solve against I
// (linearization_power_varproj.hpp:141-154).  Sixteen lanes per camera, four cameras per 64-thread workgroup
// (one thread per camera was a 1 900-step dependent chain through LDS: 31 us for 1 778 cameras).
constexpr int K8_THREADS = 64, K8_CAMS_PER_WG = 4;
POVAR_KERNEL __launch_bounds__(K8_THREADS) void cam_build_binv(Dp d, double lambda) {
  __shared__ double As[K8_CAMS_PER_WG][144];
  const int q = threadIdx.x >> 4, l = threadIdx.x & 15;
  const int c = blockIdx.x * K8_CAMS_PER_WG + q;
  const bool in = c < d.n_cams;
  double* A = As[q];
  if (in) {
    const double* g = d.G + 40 * (size_t)c;
    const double* sg = d.sigma + 12 * (size_t)c;
    const double sb2 = d.sb * d.sb;
    for (int e = l; e < 144; e += 16) {
      const int r = e / 12, cc = e % 12, a = r >> 2, i = r & 3, b = cc >> 2, j = cc & 3;
      const int ij = sym10(i, j);
      double v;
      if (a == b) v = a < 2 ? g[ij] : sb2 * g[30 + ij];
      else if (a + b == 1) v = 0;
      else {
        const int k = (a == 2 ? b : a);  // 0 -> u moment, 1 -> v moment
        v = -sb2 * g[10 * (k + 1) + ij];
      }
      v = v * sg[r] * sg[cc];
      A[e] = r == cc ? v + lambda : v;
    }
  } else {
    for (int e = l; e < 144; e += 16) A[e] = (e / 12 == e % 12) ? 1.0 : 0.0;
  }
  __syncthreads();
  chol_inverse_16<12>(A, l, in ? d.binv + 144 * (size_t)c : nullptr);
}

// z_c = sigma * x_c goes to the dense vector and, for a cached camera, into the contiguous record
// image the E0 kernels copy into LDS (so their prologue is a coalesced copy, not a gather)
__device__ inline void store_z(const Dp& d, int c, int j, double v) {
  d.z[12 * (size_t)c + j] = v;
  const int r = d.cam_hot[c];
  if (r > 0) {
    d.hot_rec[(size_t)(r - 1) * HOT_REC_STRIDE + j] = v;
    if (d.zimg) d.zimg[(size_t)(r - 1) * 12 + j] = v;  // the z image e0_ck gathers from (povar_kernels_ck.hpp: ck_load_z_img)
  }
}

// static part of the hot camera records (per linearisation): P[:, :3] row-major (step 1, hom = 0)
// or the full P (step 2, hom = 1) after the 12 z values
POVAR_KERNEL __launch_bounds__(256) void build_hot_rec(Dp d, int hom) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.n_cams * 12) return;  // every camera: the image is in popularity order (Dp::hot_cams)
  const int r = i / 12, e = i % 12;
  const double* P = reinterpret_cast<const double*>(d.cams_lin4) + 12 * (size_t)d.hot_cams[r];
  double v = 0;
  if (hom) v = P[e];
  else if (e < 9) v = P[(e / 3) * 4 + (e % 3)];
  else v = P[(e - 9) * 4 + 3];  // the translation column: prepare_lpl rebuilds the residual from the full P
  d.hot_rec[(size_t)r * HOT_REC_STRIDE + 12 + e] = v;
}

// fixed-order sum of a camera's scatter items (+ the LDS-accumulated workgroup partials of a cached
// camera): lanes stride over the parts, then a butterfly; every lane ends with the 12 sums
__device__ inline void camera_item_sum(const Dp& d, int c, int lane, double (&y)[12]) {
#pragma unroll
  for (int j = 0; j < 12; ++j) y[j] = 0;
  for (int it = d.cmv.cam_item_off[c] + lane; it < d.cmv.cam_item_off[c + 1]; it += WAVE) {
    const double* ip = d.cmv.part + 12 * (size_t)it;
#pragma unroll
    for (int j = 0; j < 12; ++j) y[j] += ip[j];
  }
  if (d.hot_part && d.part_range) {
    const int2 rr = d.part_range[c];
    for (int w = rr.x + lane; w < rr.y; w += WAVE) {
      const double* ip = d.hot_part + (size_t)w * 12;
#pragma unroll
      for (int j = 0; j < 12; ++j) y[j] += ip[j];
    }
  } else if (d.hot_part) {
    const int r = d.cam_hot[c];
    if (r > 0 && r <= d.n_hot_acc) {
      for (int w = lane; w < d.n_hot_wg; w += WAVE) {
        const double* ip = d.hot_part + ((size_t)(r - 1) * d.n_hot_wg + w) * 12;
#pragma unroll
        for (int j = 0; j < 12; ++j) y[j] += ip[j];
      }
    }
  }
  wave_sum<12>(y);
}

// LDSACC modes: y_c = sigma * ( sum over the camera's COLD observations of (h q0; h q1; h q2)
//                                + sum of the workgroups' LDS-accumulated partials of a cached camera ).
constexpr int CCS_THREADS = 128;  // threads per camera of the per-camera kernels of the term loop (cam_cold_sum[_binv][_h])
// One CCS_THREADS-thread workgroup per camera, fixed summation order; replaces cm_scatter + the item sums
// (a single wavefront walking a few hundred items per camera was a serial chain of dependent loads).
template <int NT>
__global__ __launch_bounds__(NT) void cam_cold_sum(Dp d, int hom) {
  const int done = d.flags[1];  // tested after the first batch of loads is in flight
  __shared__ double sh[4 * 12];
  const int c = blockIdx.x, t = threadIdx.x;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0;
  const int2 pr = d.cmv.cam_range[c];
  const int p0 = pr.x, p1 = pr.y;
  const int r = d.hot_part ? d.cam_hot[c] : 0;
  const double sg_t = t < 12 ? d.sigma[12 * (size_t)c + t] : 0.0;  // requested early, used last
  if (done) return;
  // 4 observations per thread in flight: index loads, then the dependent gathers, then the FMAs
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
      hw[u] = hom ? d.cmv.h[3 * d.cmv.n + pc] : 1.0;
      q[u] = in ? d.q4c[d.cmv.src ? d.cmv.src[pc] : pc] : make_double4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] += hx[u] * q[u].x; acc[1] += hy[u] * q[u].x; acc[2] += hz[u] * q[u].x; acc[3] += hw[u] * q[u].x;
      acc[4] += hx[u] * q[u].y; acc[5] += hy[u] * q[u].y; acc[6] += hz[u] * q[u].y; acc[7] += hw[u] * q[u].y;
      acc[8] += hx[u] * q[u].z; acc[9] += hy[u] * q[u].z; acc[10] += hz[u] * q[u].z; acc[11] += hw[u] * q[u].z;
    }
  }
  if (d.part_range) {  // e0_lpl: the camera's partial records are one contiguous run
    const int2 rr = d.part_range[c];
    for (int w = rr.x + t; w < rr.y; w += NT) {
      const double* ip = d.hot_part + (size_t)w * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] += ip[k];
    }
  } else if (r > 0 && r <= d.n_hot_acc) {
    for (int w = t; w < d.n_hot_wg; w += NT) {
      const double* ip = d.hot_part + ((size_t)(r - 1) * d.n_hot_wg + w) * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] += ip[k];
    }
  }
  block_sum_dpp<12, NT>(acc, sh);
  if (t < 12) {
    double v = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) v = (t == k) ? acc[k] : v;
    v *= sg_t;
    if (d.p2p_peer) {
      // push this rank's partial of camera c into the slab [parity][rank] of EVERY rank's exchange buffer, then
      // publish it with the epoch tag in the record's 13th entry.  Every store and every load of these bytes is a
      // system-scope (sc0 sc1, write-through / cache-bypassing) access and the storing wavefront drains its stores
      // (s_waitcnt vmcnt(0)) before the tag: no release fence -- a system-scope fence writes the whole L2 back,
      // 44 us per term with one per camera (MI355X_MICROARCH.md, "Valid forms").  Lanes 0..11 are one wavefront.
      const unsigned long long ep = *d.p2p_epoch;
      const size_t off = ((((size_t)(ep & 1) * d.p2p_world + d.p2p_rank) * d.n_cams) + c) * 16;
      for (int p = 0; p < d.p2p_world; ++p)
        __hip_atomic_store(d.p2p_peer[p] + off + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (t == 0)
        for (int p = 0; p < d.p2p_world; ++p)
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(d.p2p_peer[p] + off + 12), ep, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      d.y[12 * (size_t)c + t] = v;
    }
  }
}

// cam_cold_sum fused with cam_binv_axpy (mode 2) for the unsharded LDSACC term loop: the workgroup that
// has just summed camera c's E0 row applies B_c^-1, the AXPY and the sigma scaling itself, so the term
// needs one kernel less (the dense y is never materialised).  Norm partials are per camera
// (series_check then sums n_cams entries).
template <int NT>
__global__ __launch_bounds__(NT) void cam_cold_sum_binv(Dp d, int want_norms) {
  const int done = d.flags[1];  // tested after the first batch of loads is in flight
  __shared__ double sh[4 * 12];
  const int c = blockIdx.x, t = threadIdx.x;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0;
  const int2 pr = d.cmv.cam_range[c];  // one load instead of the two-level item index
  const int2 rr = d.part_range ? d.part_range[c] : make_int2(0, 0);  // requested with it: the partial loop does not wait a round trip of its own
  const int p0 = pr.x, p1 = pr.y;
  const int r = d.hot_part ? d.cam_hot[c] : 0;
  // everything the tail needs that depends on c only is requested now, off the critical path
  const size_t base = 12 * (size_t)c;
  double bi[12], sg[12], acc_old = 0;
  if (t < 12) {
    const double* Bi = d.binv + 144 * (size_t)c + 12 * t;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      bi[j] = Bi[j];
      sg[j] = d.sigma[base + j];
    }
    acc_old = d.accum[base + t];
  }
  if (done) return;
  constexpr int U = 4;  // (8 with the gather through CmView::src: 1499 -> 1470 terms/s on final-13682)
  for (int pb = p0 + t; pb < p1; pb += U * NT) {
    double hx[U], hy[U], hz[U];
    double4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = pb + u * NT;
      const bool in = p < p1;
      const int pc = in ? p : p0;
      hx[u] = d.cmv.h[pc];
      hy[u] = d.cmv.h[d.cmv.n + pc];
      hz[u] = d.cmv.h[2 * d.cmv.n + pc];
      q[u] = in ? d.q4c[d.cmv.src ? d.cmv.src[pc] : pc] : make_double4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] += hx[u] * q[u].x; acc[1] += hy[u] * q[u].x; acc[2] += hz[u] * q[u].x; acc[3] += q[u].x;
      acc[4] += hx[u] * q[u].y; acc[5] += hy[u] * q[u].y; acc[6] += hz[u] * q[u].y; acc[7] += q[u].y;
      acc[8] += hx[u] * q[u].z; acc[9] += hy[u] * q[u].z; acc[10] += hz[u] * q[u].z; acc[11] += q[u].z;
    }
  }
  if (d.part_range) {  // e0_lpl: the camera's partial records are one contiguous run
    for (int w = rr.x + t; w < rr.y; w += NT) {
      const double* ip = d.hot_part + (size_t)w * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] += ip[k];
    }
  } else if (r > 0 && r <= d.n_hot_acc) {
    for (int w = t; w < d.n_hot_wg; w += NT) {
      const double* ip = d.hot_part + ((size_t)(r - 1) * d.n_hot_wg + w) * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] += ip[k];
    }
  }
  block_sum_dpp<12, NT>(acc, sh);  // every thread now holds the 12 sums
  if (t >= 64) return;
  double nrm[2] = {0, 0};
  if (t < 12) {
    const size_t idx = base + t;
    double s = 0, sgt = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      s += bi[j] * (acc[j] * sg[j]);
      sgt = (t == j) ? sg[j] : sgt;
    }
    const double a = acc_old + s;
    d.tmp[idx] = s;
    d.accum[idx] = a;
    store_z(d, c, t, s * sgt);
    nrm[0] = s * s;
    nrm[1] = a * a;
  }
  if (want_norms) {
    wave_sum<2>(nrm);
    if (t == 0) {
      d.norm_part[2 * (size_t)c] = nrm[0];
      d.norm_part[2 * (size_t)c + 1] = nrm[1];
    }
  }
}

// b_c = sigma * sum_items (scatter parts)   (landmark_block.hpp:529-534); one wavefront per camera
POVAR_KERNEL __launch_bounds__(256) void cam_sum_items(Dp d, double* out, int apply_sigma) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= d.n_cams) return;
  double y[12];
  camera_item_sum(d, c, lane, y);
  if (lane < 12) {
    double v = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) v = (lane == j) ? y[j] : v;
    out[12 * (size_t)c + lane] = apply_sigma ? v * d.sigma[12 * (size_t)c + lane] : v;
  }
}

// K9 + K11: tmp = B^-1 y, accum (+)= tmp, z = sigma * tmp, optional squared-norm partials
// (right_mul_b_inv_pOSE + the loop body of solve_pOSE, linearization_power_varproj.hpp:196-207,
// 322-340).  mode 0: y = -b (series start); 1: y = sigma * sum of scatter items (implicit E0);
// 2: y = dense buffer d.y (the per-camera sums of the LDSACC modes, or the all-reduced vector).
constexpr int K9_CAMS = 4;  // one wavefront per camera, 4 cameras per workgroup
POVAR_KERNEL __launch_bounds__(K9_CAMS * 64) void cam_binv_axpy(Dp d, int mode, int want_norms) {
  const int done = mode != 0 ? d.flags[1] : 0;  // tested before the first store: its round trip overlaps the loads
  __shared__ double sh[K9_CAMS * 2];
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * K9_CAMS + (threadIdx.x >> 6);
  const bool in = c < d.n_cams;
  double y[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) y[j] = 0;
  if (in) {
    const size_t base = 12 * (size_t)c;
    if (mode == 0) {
#pragma unroll
      for (int j = 0; j < 12; ++j) y[j] = -d.b[base + j];
    } else if (mode == 1) {
      camera_item_sum(d, c, lane, y);
#pragma unroll
      for (int j = 0; j < 12; ++j) y[j] *= d.sigma[base + j];
    } else if (mode == 5) {
      // peer-to-peer exchange: wait for every rank's slab of this camera (tag == epoch), sum in rank order
      const unsigned long long ep = *d.p2p_epoch;
      const double* mine = d.p2p_peer[d.p2p_rank];
      const bool gave_up = (d.flags[0] & 2) != 0;  // an earlier wait of this solve timed out: do not wait again
      for (int p = 0; p < d.p2p_world; ++p) {
        const double* rec = mine + ((((size_t)(ep & 1) * d.p2p_world + p) * d.n_cams) + c) * 16;
        int spins = 0;
        while (!gave_up && __hip_atomic_load(reinterpret_cast<const unsigned long long*>(rec + 12), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM) != ep) {  // relaxed poll; the data loads below bypass the caches too
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 22)) {  // a peer never arrived: flag it and go on (the host reports the failure)
            if (lane == 0) atomicOr(&d.flags[0], 2);
            break;
          }
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) y[j] += __hip_atomic_load(rec + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) y[j] = d.y[base + j];
    }
  }
  double nrm[2] = {0, 0};
  double s = 0, acc = 0, sg = 0;
  if (in && lane < 12) {
    const size_t idx = 12 * (size_t)c + lane;
    const double* Bi = d.binv + 144 * (size_t)c + 12 * lane;
#pragma unroll
    for (int j = 0; j < 12; ++j) s += Bi[j] * y[j];
    acc = mode == 0 ? s : d.accum[idx] + s;
    sg = d.sigma[idx];
  }
  if (done) return;
  if (in && lane < 12) {
    const size_t idx = 12 * (size_t)c + lane;
    d.tmp[idx] = s;
    d.accum[idx] = acc;
    store_z(d, c, lane, s * sg);
    nrm[0] = s * s;
    nrm[1] = acc * acc;
  }
  if (want_norms) {
    block_sum<2, K9_CAMS * 64>(nrm, sh);
    if (threadIdx.x == 0) {
      d.norm_part[2 * (size_t)blockIdx.x] = nrm[0];
      d.norm_part[2 * (size_t)blockIdx.x + 1] = nrm[1];
    }
  }
}

// convergence tests of solve_pOSE (linearization_power_varproj.hpp:198, 206-229), on the device
// so the m-term loop needs no host round trip; later kernels of the loop see flags[1] and exit.
POVAR_KERNEL __launch_bounds__(64) void series_check(Dp d, int n_blocks, int i, double q_tol, double r_tol) {
  if (d.flags[1]) return;
  double v[2] = {0, 0};
  for (int k = threadIdx.x; k < n_blocks; k += 64) {
    v[0] += d.norm_part[2 * (size_t)k];
    v[1] += d.norm_part[2 * (size_t)k + 1];
  }
  wave_sum<2>(v);
  if (threadIdx.x != 0) return;
  const double iter_norm = sqrt(v[0]), acc_norm = sqrt(v[1]);
  if (i == 0) {
    d.norms[0] = acc_norm;
    return;
  }
  d.norms[1] = iter_norm;
  d.norms[2] = acc_norm;
  bool conv = false;
  if (q_tol > 0 && i * iter_norm / acc_norm < q_tol) conv = true;
  if (!conv && r_tol > 0 && iter_norm / d.norms[0] < r_tol) conv = true;
  if (conv) {
    d.flags[1] = 1;
    d.flags[2] = i;
    d.flags[3] = 1;
  }
}

// K13: P_c += reshape(inc * sigma) and the scale / unscale round trip of
// linearizor_power_varproj.cpp:251-255.  mode 0 (VARPROJ): cams += inc*sigma, inc <- (inc*sigma)*(1/sigma)
// mode 1 (POWER_SCHUR_COMPLEMENT, before back substitution): z = sigma*inc only
// mode 2 (POWER_SCHUR_COMPLEMENT, after): cams += inc*sigma
POVAR_KERNEL __launch_bounds__(256) void cam_apply_inc(Dp d, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 12 * d.n_cams) return;
  const double sg = d.sigma[i];
  const double s = d.inc[i] * sg;
  double* cams = reinterpret_cast<double*>(d.cams4);
  if (mode == 0) {
    cams[i] += s;
    d.inc[i] = s * (1.0 / sg);
  } else if (mode == 1) {
    store_z(d, i / 12, i % 12, s);
  } else {
    cams[i] += s;
  }
}

// lane-ordered mirror of a per-landmark array (V2::lmx / lml / lsc): out[tile][lane] = in[landmark of the lane]
POVAR_KERNEL __launch_bounds__(256) void lm_to_lanes(const int* lm_of, const double4* in, double4* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int lm = lm_of[i];
  out[i] = lm >= 0 ? in[lm] : make_double4(0, 0, 0, 0);
}
// and back: the landmark-order master of a lane-ordered array (first lane of each landmark)
POVAR_KERNEL __launch_bounds__(256) void lanes_to_lm(const int* lm_of, const int* seg, const double4* in, double4* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int lm = lm_of[i];
  if (lm >= 0 && (int)(i & 63) == (seg[i] & 255)) out[lm] = in[i];
}
POVAR_KERNEL __launch_bounds__(256) void lms3_to_4(const double* in, double4* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = make_double4(in[3 * i], in[3 * i + 1], in[3 * i + 2], 1.0);
}
POVAR_KERNEL __launch_bounds__(256) void lms4_to_3(const double4* in, double* out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const double4 v = in[i];
    out[3 * i] = v.x;
    out[3 * i + 1] = v.y;
    out[3 * i + 2] = v.z;
  }
}

}  // namespace povar
