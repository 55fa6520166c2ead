// povar_kernels_res.hpp -- gfx950 device code of the RESIDENT power series: the whole loop of solve_pOSE
// (linearization_power_varproj.hpp:191-237: x_0 = B^-1 (-b); x_i = B^-1 E0 x_{i-1}; sum, early exit) in ONE launch, for
// contexts whose term-invariant operands fit on the chip (res_layout.hpp).
//
// The per-term kernels re-read per term what does not change between the terms of a solve: rows, landmark records, tile
// metadata, B^-1 -- and pay a launch ramp, a flush and two kernel boundaries each time (29 us per term on a venice-1778
// shard of one rank in eight, where the rows cost 7; profiles/r04_shard_term_times.txt).  Here
//   * a LANE keeps its chunks of <= H observations of one camera each (uv, landmark slot, robust weight) and that camera's
//     P3 = P_c[:, :3] in REGISTERS for the whole solve;
//   * the workgroup's LANDMARKS live in LDS (h~ 24 bytes, u / g 24 bytes; G = diag(s) Hll^-1 diag(s) in the registers of
//     the lane that owns the slot);
//   * every camera has an OWNER workgroup that holds B_c^-1, sigma_c, the running sum and the last term in LDS.
// Per term only z = sigma x and the per-camera partial sums move (right_mul_e0_pOSE :364-406 + right_mul_b_inv_pOSE
// :322-340), through ONE region of LDS that is in turn
//   1. the z of the workgroup's cameras (gathered from the z table by all threads, one 16-byte granule pair each),
//   2. their accumulators: forward u_l += P3^T (w C (Z h~)) (LDS atomics on the landmark), g = G u, backward
//      y_c += h~ (x) (w C (P3 g)) (registers, one segmented wavefront sum, one LDS add per run of lanes),
//   3. after the accumulators have gone out as the workgroup's partial records (contiguous: a coalesced copy): the
//      records the workgroup reads as an OWNER -- all threads again, a granule pair each --, which it sums camera by
//      camera in a fixed order, applies B_c^-1 to, adds to the sum and publishes as the next z_c.
// The reference's mutex-guarded `res += Jp^T s` (:388-398) is the LDS accumulation + the owner's fixed-order sum.
// (A first version gave every lane its camera's z and its partial record to itself: twelve 16-byte accesses per chunk
// and hand-over, each wavefront instruction touching 64 lines -- the address pipes of the CUs, not a latency, set the
// term at 12 us on ladybug-49 and 44 us on a venice shard; profiles/r05_res_experiments.txt.)
//
// Hand-overs (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", R2: "the data IS the
// flag"): every handed-over double travels as TWO 8-byte granules {32 bits of the value, tag} written by one 16-byte
// agent-scope (sc1: write-through, L1-bypassing) store; tag = launch number << 8 | term, so a granule of an earlier term or
// an earlier launch never matches.  The reader re-reads its granules (16-byte sc1 loads) until every tag is the one it
// waits for: no flag word, no fence, no drain of the producer's stores, no dependence on dispatch order or placement; a
// granule is written by ONE store and is never torn.  What may be overwritten when:
//   * z_c of term i replaces z_c of term i - 1 after the owner has seen ALL records of c of term i, i.e. after every
//     workgroup that reads z_c has used it;
//   * workgroup w's record of camera c of term i + 1 replaces that of term i after w has seen z_c of term i, which the
//     owner publishes after it has read the record.
//   * the norm granules of a workgroup (early-exit tests only) are DOUBLE-BUFFERED by the parity of the term their tag
//     names: workgroup g publishes the norms tagged i + 1 once it has z tag i and the records of its cameras -- which does not
//     wait for a workgroup s that contributes to none of g's cameras and may still be sweeping the norms tagged i.  With one
//     buffer s would find tag i + 1 in g's slot, never match and spin out (the resident path lost, silently: ADVICE r05).
//     With two, g cannot write tag i + 2 into the half of tag i before it has passed its own sweep of the norms tagged
//     i + 1, which needs s's granule tagged i + 1, which s publishes only after its sweep of tag i.
// Every spin is bounded (ResP::spin_limit): a launch whose workgroups are not all resident (another context's kernels on
// the device) gives up, raises bit 2 of flags[0] and the library repeats the solve with the per-term kernels.
#pragma once

#include "povar_kernels_ck.hpp"
#include "res_layout.hpp"

namespace povar {

struct ResP {
  const int* lane_cam;    // [W][R][T] camera slot of the chunk
  const int* lane_seg;
  const double2* uv;      // [W][R][H][T]
  const int* lslot;
  const int* oslot;
  const int* wave_h;      // [W][R][NW]
  const int* lm_off;      // [W + 1]
  const int* lm_id;
  const int* cam_off;     // [W + 1] camera slots = partial records of the workgroup
  const int* cam_id;
  const int* cam_zi;
  const int* own_off;     // [W + 1]
  const int* own_cam;
  const int* own_zi;
  const int2* own_q;
  const int* oq_off;      // [W + 1]
  const int* oq_rec;
  uint4* part;            // [n_rec][12] partial records, workgroup-major, one 16-byte granule pair per entry
  uint4* zbuf;            // [n_cams][12] z = sigma x of the current term, rows in popularity order
  uint4* nrm;             // [2][RES_MAX_WG][2] squared norms of (term, sum) over the cameras a workgroup owns, by term parity
  unsigned* launch;       // launch counter (the high bits of the tags)
  unsigned part_bytes, z_bytes, nrm_bytes;
  int W, m, want_norms, want_norm0;
  int w_mode;             // robust weight of an observation: 1: V2::w through V2::of_slot, 2: Dp::sw squared
  double q_tol, r_tol;
  unsigned spin_limit;
};

typedef unsigned __attribute__((ext_vector_type(4))) res_u4;
struct ResBufs {
  __amdgpu_buffer_rsrc_t part, z, nrm;
};
__device__ inline ResBufs res_bufs(const ResP& k) {
  ResBufs B;
  B.part = __builtin_amdgcn_make_buffer_rsrc(k.part, 0, k.part_bytes, 0x00020000);
  B.z = __builtin_amdgcn_make_buffer_rsrc(k.zbuf, 0, k.z_bytes, 0x00020000);
  B.nrm = __builtin_amdgcn_make_buffer_rsrc(k.nrm, 0, k.nrm_bytes, 0x00020000);
  return B;
}
constexpr int RES_SC1 = 16;  // aux bits of the buffer instructions: sc1 (agent scope)
// byte offset of norm granule pair e (0: term, 1: sum) of workgroup w in the half of the tag's parity
__device__ inline unsigned res_nrm_off(unsigned tag, int w, int e) { return ((tag & 1u) * (unsigned)(2 * RES_MAX_WG) + (unsigned)(2 * w + e)) * 16u; }
// one double as a granule pair {lo, tag, hi, tag}
__device__ inline void res_put(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double v, unsigned tag) {
  res_u4 g;
  g.x = (unsigned)__double2loint(v); g.y = tag; g.z = (unsigned)__double2hiint(v); g.w = tag;
  __builtin_amdgcn_raw_buffer_store_b128(g, r, byte_off, 0, RES_SC1);
}
// N granule pairs (one per byte offset; inactive entries are not loaded), re-read until every tag matches for every lane
// of the wavefront; false: the spin budget ran out
template <int N>
__device__ inline bool res_get(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[N], const bool (&active)[N], unsigned tag, double (&v)[N],
                               unsigned limit) {
  for (unsigned spins = 0;; ++spins) {
    res_u4 g[N];
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] = __builtin_amdgcn_raw_buffer_load_b128(r, active[e] ? off[e] : 0u, 0, RES_SC1);
    bool ok = true;
#pragma unroll
    for (int e = 0; e < N; ++e) ok = ok && (!active[e] || (g[e].y == tag && g[e].w == tag));
    if (__all(ok)) {
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] = __hiloint2double((int)g[e].z, (int)g[e].x);
      return true;
    }
    if (spins >= limit) return false;
    __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");  // (the loads above are re-issued: the compiler must not keep their values)
  }
}

// All threads of the workgroup: granule pair i of a list of n (pair i is entry i % 12 of row row_of(i / 12) of the
// buffer) -> dst[(i / 12) * stride + i % 12], RES_GB pairs per thread in flight (all of a workgroup's in one round trip
// where the registers allow: 1024-thread workgroups have few cameras each)
template <int T, int RES_GB, class RowOf>
__device__ inline bool res_gather(__amdgpu_buffer_rsrc_t r, int n, RowOf row_of, unsigned tag, double* dst, int stride, int t, unsigned limit) {
  bool fine = true;
  for (int i0 = 0; i0 < n; i0 += RES_GB * T) {
    if (i0 + (t & ~63) >= n) break;  // (wave-uniform: nothing left for this wavefront)
    unsigned off[RES_GB];
    bool act[RES_GB];
    double v[RES_GB];
#pragma unroll
    for (int b = 0; b < RES_GB; ++b) {
      const int i = i0 + b * T + t;
      act[b] = i < n;
      off[b] = act[b] ? ((unsigned)row_of(i / 12) * 12u + (unsigned)(i % 12)) * 16u : 0u;
    }
    fine = res_get<RES_GB>(r, off, act, tag, v, limit) && fine;
#pragma unroll
    for (int b = 0; b < RES_GB; ++b) {
      const int i = i0 + b * T + t;
      if (act[b]) dst[(i / 12) * stride + i % 12] = v[b];
    }
  }
  return fine;
}

// the per-lane state of one chunk
template <int H>
struct ResChunk {
  double2 uv[H];
  int ls[H];
  double rw[H];
  double P3[9];
  int ci, seg, hrows, dup, steps;
};

// NW wavefronts per workgroup, chunks of H rows, RR chunks per lane, LS landmark slots per lane
template <int NW, int H, int RR, int LS, bool ROBUST>
__global__ __launch_bounds__(NW * 64) void series_res(Dp d, ResP k) {
  constexpr int T = NW * 64;
  constexpr int GB = NW >= 16 ? 4 : 8;
  extern __shared__ double res_lds[];
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const ResBufs B = res_bufs(k);
  const unsigned tag0 = (*k.launch) << 8;  // (the counter is bumped by a kernel behind this one: stream order)
  const int L0 = k.lm_off[g], nL = k.lm_off[g + 1] - L0;
  const int C0 = k.cam_off[g], nC = k.cam_off[g + 1] - C0;
  const int O0 = k.own_off[g], nO = k.own_off[g + 1] - O0;
  const int Q0 = k.oq_off[g], nQ = k.oq_off[g + 1] - Q0;
  int* ctl = reinterpret_cast<int*>(res_lds);  // [0] a wait gave up, [1] series converged, [2] iterations, [4..5] |x_0|
  double* lh = res_lds + 8;             // [nL][3] landmark coordinates
  double* lu = lh + 3 * nL;             // [nL][3] u = Jl^T Jp x, then g = G u
  double* reg = lu + 3 * nL;            // the region: z of the cameras [nC][13] -> accumulators [nC][13] -> owner's records [nQ][12]
  double* obinv = reg + res_region_doubles(nC, nQ);  // [nO][144] B^-1 of the owned cameras
  double* osig = obinv + 144 * nO;      // [nO][12] sigma
  double* oacc = osig + 12 * nO;        // [nO][12] running sum
  double* otmp = oacc + 12 * nO;        // [nO][12] last term
  double* oy = otmp + 12 * nO;          // [nO][12] E0 row of the term: sigma * sum of the camera's records
  double* onrm = oy + 12 * nO;          // [nO][2] squared norms of the last term / the sum
  double* ops = onrm + 2 * nO;          // [nO][5][12] partial sums of the camera's records (five groups of twelve lanes)
  int* lzi = reinterpret_cast<int*>(ops + 60 * nO);  // [nC] z-table row of each camera slot
  int* loq = lzi + nC;                  // [nQ] the records read as an owner
  int* lown = loq + nQ;                 // [nO][4] z-table row, first / end position of the records

  // ---------------- prologue: everything that does not change between the terms
  // owned cameras first (their registers are free again before the lane's own state is loaded): B^-1, sigma;
  // x_0 = B^-1 (-b) (the series start, :196); z_0 published
  if (t < 4) ctl[t] = 0;
  for (int e = t; e < nC; e += T) lzi[e] = k.cam_zi[C0 + e];
  for (int e = t; e < nQ; e += T) loq[e] = k.oq_rec[Q0 + e];
  for (int e = t; e < nO; e += T) {
    const int2 qr = k.own_q[O0 + e];
    lown[4 * e] = k.own_zi[O0 + e]; lown[4 * e + 1] = qr.x; lown[4 * e + 2] = qr.y;
  }
  for (int o = wave; o < nO; o += NW) {
    const int c = k.own_cam[O0 + o];
    for (int e = lane; e < 144; e += 64) obinv[144 * o + e] = d.binv[144 * (size_t)c + e];
    if (lane < 12) osig[12 * o + lane] = d.sigma[12 * (size_t)c + lane];
    if (lane >= 32 && lane < 44) otmp[12 * o + lane - 32] = -d.b[12 * (size_t)c + lane - 32];
  }
  __syncthreads();
  for (int o = wave; o < nO; o += NW) {
    const int zi = lown[4 * o];
    double s = 0;
    if (lane < 12) {
      const double* Bi = obinv + 144 * o + 12 * lane;
#pragma unroll
      for (int j = 0; j < 12; ++j) s += Bi[j] * otmp[12 * o + j];
      oacc[12 * o + lane] = s;
      res_put(B.z, (unsigned)(12 * zi + lane) * 16u, s * osig[12 * o + lane], tag0 | 1u);
    }
    if (k.want_norm0) {
      double n2[1] = {s * s};
      wave_sum<1>(n2);
      if (lane == 0) { onrm[2 * o] = n2[0]; onrm[2 * o + 1] = n2[0]; }
    }
    // (otmp = x_0 once every lane of the wavefront has read -b from it)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane < 12) otmp[12 * o + lane] = s;
  }
  if (k.want_norm0) {
    __syncthreads();
    if (t == 0) {
      double a = 0;
      for (int o = 0; o < nO; ++o) a += onrm[2 * o];
      res_put(B.nrm, res_nrm_off(tag0 | 1u, g, 0), a, tag0 | 1u);
      res_put(B.nrm, res_nrm_off(tag0 | 1u, g, 1), a, tag0 | 1u);
    }
  }
  // the lane's chunks: rows, camera slot, P3
  ResChunk<H> ch[RR];
#pragma unroll
  for (int r = 0; r < RR; ++r) {
    const size_t li = ((size_t)g * RR + r) * T + t;
    ch[r].ci = k.lane_cam[li];
    ch[r].seg = k.lane_seg[li];
    const int wh = __builtin_amdgcn_readfirstlane(k.wave_h[((size_t)g * RR + r) * NW + wave]);
    ch[r].hrows = wh & 255;
    ch[r].dup = (wh >> 8) & 1;
    ch[r].steps = (wh >> 12) & 15;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const size_t row = (((size_t)g * RR + r) * H + j) * T + t;
      ch[r].uv[j] = k.uv[row];
      ch[r].ls[j] = k.lslot[row];
      ch[r].rw[j] = 1.0;
      if (ROBUST) {
        const int os = k.oslot[row];
        if (os >= 0) {
          if (k.w_mode == 1) ch[r].rw[j] = d.v2.w[d.v2.of_slot[os]];
          else { const double s = d.sw[os]; ch[r].rw[j] = s * s; }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) ch[r].P3[e] = 0;
    if (ch[r].ci >= 0) {
      const Cam P = load_cam(d.cams_lin4, k.cam_id[C0 + ch[r].ci]);
      ch[r].P3[0] = P.r0.x; ch[r].P3[1] = P.r0.y; ch[r].P3[2] = P.r0.z;
      ch[r].P3[3] = P.r1.x; ch[r].P3[4] = P.r1.y; ch[r].P3[5] = P.r1.z;
      ch[r].P3[6] = P.r2.x; ch[r].P3[7] = P.r2.y; ch[r].P3[8] = P.r2.z;
    }
  }
  // the landmark slots of the lane: h~ into LDS, u = 0, G = diag(s) Hll^-1 diag(s) in registers
  double G[LS][6];
#pragma unroll
  for (int q = 0; q < LS; ++q) {
#pragma unroll
    for (int e = 0; e < 6; ++e) G[q][e] = 0;
    const int s = t + q * T;
    if (s < nL) {
      const int lm = k.lm_id[L0 + s];
      const int pos = d.v2.lm_pos[lm] & ((1 << 26) - 1);
      const double* rp = d.v2.lmrec + ((size_t)(pos >> 6) * 9) * WAVE + (pos & 63);
      lh[3 * s] = rp[0];
      lh[3 * s + 1] = rp[WAVE];
      lh[3 * s + 2] = rp[2 * WAVE];
      lu[3 * s] = 0; lu[3 * s + 1] = 0; lu[3 * s + 2] = 0;
#pragma unroll
      for (int e = 0; e < 6; ++e) G[q][e] = rp[(3 + e) * WAVE];
    }
  }
  int iters = k.m;
  auto z_row = [&](int s) { return lzi[s]; };
  auto q_row = [&](int q) { return loq[q]; };

  // ---------------- the terms
  for (int i = 1; i <= k.m + 1; ++i) {
    const unsigned tag = tag0 | (unsigned)i;
    if (i == k.m + 1 && !k.want_norms) break;  // (with the tests on: the last term's norms are looked at too)
    // ---- the norms of term i - 1 (every owner's: the one step of a term that waits for ALL workgroups)
    if (k.want_norms && (i > 1 || k.want_norm0)) {
      if (wave == 0) {
        double v[2] = {0, 0};
        bool ok = true;
        for (int w0 = 0; w0 < k.W; w0 += 64) {
          const unsigned off[2] = {res_nrm_off(tag, w0 + lane, 0), res_nrm_off(tag, w0 + lane, 1)};
          const bool act[2] = {w0 + lane < k.W, w0 + lane < k.W};
          double e2[2];
          ok = res_get<2>(B.nrm, off, act, tag, e2, k.spin_limit) && ok;
          if (act[0]) { v[0] += e2[0]; v[1] += e2[1]; }
        }
        if (!ok && lane == 0) ctl[0] = 1;
        wave_sum<2>(v);
        const double iter_norm = sqrt(v[0]), acc_norm = sqrt(v[1]);
        if (i == 1) {
          if (lane == 0) reinterpret_cast<double*>(ctl)[2] = acc_norm;
          if (g == 0 && lane == 0) d.norms[0] = acc_norm;
        } else {
          const double n0 = reinterpret_cast<double*>(ctl)[2];
          bool conv = false;
          if (k.q_tol > 0 && (i - 1) * iter_norm / acc_norm < k.q_tol) conv = true;    // :206-214
          if (!conv && k.r_tol > 0 && iter_norm / n0 < k.r_tol) conv = true;             // :216-229
          if (ok && conv && lane == 0) { ctl[1] = 1; ctl[2] = i - 1; }
          if (g == 0 && lane == 0) { d.norms[1] = iter_norm; d.norms[2] = acc_norm; }
        }
      }
      __syncthreads();
      if (ctl[0] | ctl[1]) break;
    }
    if (i == k.m + 1) break;
    // ---- hand-over 2: z of the workgroup's cameras into the region
    // (the region's last readers were the owner sums of term i - 1: the barrier in front of them is B6 below, the one
    // behind them the barrier at the end of the term)
    if (!res_gather<T, GB>(B.z, nC * 12, z_row, tag, reg, RES_ACC_STRIDE, t, k.spin_limit) && lane == 0) ctl[0] = 1;
    __syncthreads();  // B1
    if (ctl[0]) break;
    // ---- forward: u_l += P3^T (w C (Z h~_l))
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      if (ch[r].hrows == 0) continue;  // (wave-uniform)
      double zz[12];
      const double* zp = reg + (ch[r].ci < 0 ? 0 : ch[r].ci) * RES_ACC_STRIDE;
#pragma unroll
      for (int e = 0; e < 12; ++e) zz[e] = zp[e];
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (j < ch[r].hrows && ch[r].ls[j] >= 0) {
          const int s = ch[r].ls[j];
          ck_obs_forward(d, ch[r].uv[j], ch[r].rw[j], zz, ch[r].P3, lh[s], lh[s + 1], lh[s + 2], lu, 0, (uint32_t)s);
        }
      }
    }
    __syncthreads();  // B2
    // ---- g = G u per landmark slot; the region becomes the accumulators
#pragma unroll
    for (int q = 0; q < LS; ++q) {
      const int s = t + q * T;
      if (s < nL) {
        const double u0 = lu[3 * s], u1 = lu[3 * s + 1], u2 = lu[3 * s + 2];
        lu[3 * s] = G[q][0] * u0 + G[q][1] * u1 + G[q][2] * u2;
        lu[3 * s + 1] = G[q][1] * u0 + G[q][3] * u1 + G[q][4] * u2;
        lu[3 * s + 2] = G[q][2] * u0 + G[q][4] * u1 + G[q][5] * u2;
      }
    }
    for (int e = t; e < nC * RES_ACC_STRIDE; e += T) reg[e] = 0;
    __syncthreads();  // B3
    // ---- backward: y_c += h~_l (x) (w C (P3 g_l)); lanes of one camera are summed, the run's last lane adds to the
    // camera's accumulator
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      if (ch[r].hrows == 0) continue;
      double y[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (j < ch[r].hrows && ch[r].ls[j] >= 0) {
          const int s = ch[r].ls[j];
          const double gg[3] = {lu[s], lu[s + 1], lu[s + 2]};
          ck_obs_backward(d, ch[r].uv[j], ch[r].rw[j], ch[r].P3, lh[s], lh[s + 1], lh[s + 2], gg, y);
        }
      }
      // (the run's LAST lane holds its sum after the scan: no broadcast)
      if (ch[r].dup) seg_scan_steps<12>(y, lane, ch[r].seg & 255, ch[r].steps);
      if (ch[r].ci >= 0 && lane == ((ch[r].seg >> 8) & 255)) {
        double* a = reg + ch[r].ci * RES_ACC_STRIDE;
        if (ch[r].seg & (1 << 16)) {  // the camera's only run in the workgroup: a plain store
#pragma unroll
          for (int e = 0; e < 12; ++e) a[e] = y[e];
        } else {
#pragma unroll
          for (int e = 0; e < 12; ++e) __hip_atomic_fetch_add(a + e, y[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    __syncthreads();  // B4
    // ---- the workgroup's partial records: one contiguous run of granule pairs (hand-over 1, the producer's side);
    // u back to zero for the next term (its last readers were the backward pass)
    for (int e = t; e < nC * 12; e += T)
      res_put(B.part, (unsigned)(C0 * 12 + e) * 16u, reg[(e / 12) * RES_ACC_STRIDE + e % 12], tag);
#pragma unroll
    for (int q = 0; q < LS; ++q) {
      const int s = t + q * T;
      if (s < nL) { lu[3 * s] = 0; lu[3 * s + 1] = 0; lu[3 * s + 2] = 0; }
    }
    __syncthreads();  // B5 (the accumulators have been read: the region becomes the owner's records)
    // ---- owners (hand-over 1): the records of the cameras the workgroup owns into the region
    if (!res_gather<T, GB>(B.part, nQ * 12, q_row, tag, reg, 12, t, k.spin_limit) && lane == 0) ctl[0] = 1;
    __syncthreads();  // B6
    if (ctl[0]) break;
    // ---- x_i = B^-1 (sigma * sum of the records), sum += x_i, z published (:200-204, :322-340).  A camera's records are
    // summed by five groups of twelve lanes (record q of the camera by group q % 5), then the five partial sums in order
    for (int o = wave; o < nO; o += NW) {
      const int zi = lown[4 * o];
      const int2 qr = make_int2(lown[4 * o + 1], lown[4 * o + 2]);
      if (lane < 60) {
        const int e = lane % 12, grp = lane / 12;
        double a = 0;
        for (int q = qr.x + grp; q < qr.y; q += 5) a += reg[12 * q + e];
        ops[60 * o + lane] = a;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane < 12) {
        const double yl = (((ops[60 * o + lane] + ops[60 * o + 12 + lane]) + ops[60 * o + 24 + lane]) + ops[60 * o + 36 + lane]) + ops[60 * o + 48 + lane];
        oy[12 * o + lane] = yl * osig[12 * o + lane];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      double s = 0, a = 0;
      if (lane < 12) {
        const double* Bi = obinv + 144 * o + 12 * lane;
#pragma unroll
        for (int j = 0; j < 12; ++j) s += Bi[j] * oy[12 * o + j];
        a = oacc[12 * o + lane] + s;
        otmp[12 * o + lane] = s;
        oacc[12 * o + lane] = a;
        res_put(B.z, (unsigned)(12 * zi + lane) * 16u, s * osig[12 * o + lane], tag + 1u);
      }
      if (k.want_norms) {
        double n2[2] = {s * s, a * a};
        wave_sum<2>(n2);
        if (lane == 0) { onrm[2 * o] = n2[0]; onrm[2 * o + 1] = n2[1]; }
      }
    }
    __syncthreads();  // B7 (the owner sums have read the region: the next term's z may land in it)
    if (k.want_norms && t == 0) {
      double a = 0, b = 0;
      for (int o = 0; o < nO; ++o) { a += onrm[2 * o]; b += onrm[2 * o + 1]; }
      res_put(B.nrm, res_nrm_off(tag + 1u, g, 0), a, tag + 1u);
      res_put(B.nrm, res_nrm_off(tag + 1u, g, 1), b, tag + 1u);
    }
  }
  // ---------------- epilogue: sum and last term of the owned cameras, status
  __syncthreads();
  if (ctl[1]) iters = ctl[2];
  for (int e = t; e < nO * 12; e += T) {
    const int c = k.own_cam[O0 + e / 12];
    d.accum[12 * (size_t)c + e % 12] = oacc[e];
    d.tmp[12 * (size_t)c + e % 12] = otmp[e];
    store_z(d, c, e % 12, otmp[e] * osig[e]);  // z = sigma x of the last term, where the per-term kernels leave it (povar_power_series_step goes on from it)
  }
  if (t == 0) {
    if (ctl[0]) atomicOr(&d.flags[0], 4);
    if (g == 0 && ctl[1]) {
      d.flags[1] = 1;
      d.flags[2] = iters;
      d.flags[3] = 1;
    }
  }
}

// the launch counter (the high bits of the granule tags of the NEXT launch): a one-thread kernel behind series_res in the
// stream -- inside the launch itself a workgroup that is late could still be reading the old value
POVAR_KERNEL void res_bump_launch(unsigned* launch) {
  unsigned v = *launch + 1u;
  if (v >= (1u << 24)) v = 1u;  // (a tag is launch << 8 | term: 24 bits; a wrap meets granules 16 M launches old)
  *launch = v;
}

}  // namespace povar
