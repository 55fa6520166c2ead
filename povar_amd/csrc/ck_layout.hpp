// ck_layout.hpp -- host-side construction of the camera-chunk layout of the per-term E0 kernel e0_ck
// (povar_kernels_ck.hpp: struct CkP).  Pure host C++, no device code.
//
// e0_lpl (lane = landmark, camera records + accumulators in LDS) is bound by the LDS pipe: 240 bytes of record reads
// and twelve fp64 atomics per observation.  e0_ck turns the split round: lane = a CHUNK of at most CK_HMAX
// observations of ONE camera (its record Z, P3 and its accumulator y_c live in registers), and the LANDMARKS of a
// batch live in LDS (h~ 24 bytes, u / g 24 bytes): per observation 24 bytes read + three atomics on the way forward,
// 48 bytes read on the way back.  The layout is derived from the lane-per-landmark layout, which has already decided
//   * which workgroup owns which landmark and which cameras have an accumulator slot in which workgroup (the camera
//     grid / range strategy of lpl_layout.hpp: a camera's observations are concentrated in the workgroups that hold it,
//     which is what makes the chunks long),
//   * the lane of every landmark (V2::lmrec is in that order: a batch is a set of lane-per-landmark tiles, so the
//     landmark records of a batch are loaded with coalesced reads and nothing per landmark has to be permuted).
// Per workgroup the tiles are dealt round-robin over NB batches (NB: what the LDS holds); the observations of a batch
// are grouped by camera, every camera's run is cut into near-equal chunks of at most H observations (H per batch: what
// the longest-first schedule of the chunk tiles over the wavefronts likes best), chunks are sorted by (length, camera)
// and cut into tiles of 64 chunks; row j of a tile holds observation j of every chunk ([row][lane]: coalesced).
// Chunks of cameras WITHOUT an accumulator slot in the workgroup ("cold" in the lane-per-landmark layout) are ordinary
// chunks here -- their record comes from the rank-ordered image like every other -- and leave their twelve sums in a
// partial record of their own instead of the LDS accumulator: no per-observation scatter, no cold view.
#pragma once

#include <climits>
#include <cmath>
#include <cstring>

#include "lpl_layout.hpp"

namespace povar {

#ifndef POVAR_CK_HMAX_BUILD
#define POVAR_CK_HMAX_BUILD 16
#endif
constexpr int CK_HMAX = POVAR_CK_HMAX_BUILD;  // rows per chunk tile at most (observations per chunk)
constexpr int CK_LDS_BYTES = 160 * 1024;
constexpr int CK_FLAG_DUP = 1;     // lanes of the tile share accumulators: segmented wavefront sum before the flush
constexpr int CK_FLAG_COLD = 2;    // some chunk of the tile writes its own partial record
constexpr uint32_t CK_NONE = 0xffffu;  // 16-bit landmark slot of a row without observation

// Packed image points.  The observations of a BAL / data_custom file are decimal numbers with six digits behind the point
// (bal_problem.cpp:373-375 writes them with "%lf", the loader's strtod gives the nearest double: u = RN(k / 10^6) with an
// integer k) -- 16 bytes per observation that carry 2 x 31 bits.  Where EVERY image point of a layout is such a number with
// |k| < 2^31, the rows keep (k_u, k_v) as two int32 (8 bytes instead of 16: the rows are read on both walks of every term)
// and the kernel rebuilds the double with ck_unpack_uv -- int -> double, one multiplication, two fused multiply-adds, all
// correctly rounded IEEE operations, so host and device compute the same bits.  ck_pack_uv verifies bit-identity for every
// entry with exactly that sequence; one entry that does not come back (any other kind of input: the ABI takes arbitrary
// doubles) keeps the whole layout on the 16-byte rows.  Nothing is approximated either way.
constexpr double CK_UV_SCALE = 1e6, CK_UV_INV = 1e-6;
inline double ck_unpack_uv_host(int32_t k) {
  const double kd = (double)k;
  const double q0 = kd * CK_UV_INV;
  const double r = std::fma(-q0, CK_UV_SCALE, kd);
  return std::fma(r, CK_UV_INV, q0);
}
inline bool ck_pack_one(double x, int32_t& k) {
  const double t = std::nearbyint(x * CK_UV_SCALE);
  if (!(std::fabs(t) < 2147483647.0)) return false;  // (also NaN)
  k = (int32_t)t;
  const double y = ck_unpack_uv_host(k);
  uint64_t a, b;
  std::memcpy(&a, &x, 8);
  std::memcpy(&b, &y, 8);
  return a == b || (x == 0.0 && y == 0.0);  // (-0.0 is stored as 0: the operator reads u, v only through products and sums with them)
}

struct CkLayout {
  std::vector<double2> uv;       // [rows][64]
  std::vector<int2> uvp;         // [rows][64] packed image points (ck_pack_uv; then uv is released)
  bool packed = false;
  // Chunks of a camera WITHOUT an accumulator slot ("cold": almost all of one observation) are 3 % of venice's observations and
  // half of its partial records (130 919 of 257 972 x 96 bytes, written and read back every term), and most of the records on a
  // graph without hubs.  With cold_q such a lane leaves only q = w C (P3 g) (32 bytes) of each observation at the observation's
  // place in the parent layout's camera-major cold view (LplLayout::cpos), and the per-camera kernel forms h~ (x) q from the
  // view's landmark copy as it does behind e0_lpl: 32 + 56 bytes per cold observation instead of 96 + 96.
  std::vector<int> cpos;         // [rows][64] cold-view position of the entry (cold lanes only, -1 otherwise); empty without cold_q
  bool cold_q = false;
  std::vector<uint32_t> li;      // [li_rows][64] 3 x landmark slot (16 bits each; CK_NONE: none) of rows 2q | 2q+1 << 16 of a tile
  std::vector<int> src;          // [rows][64] row slot of the lane-per-landmark layout this entry is (-1: none)
  std::vector<int4> tile;        // x: first row, y: height, z: flags, w: first li row
  std::vector<int> lane_cam;     // [tiles][64] popularity rank of the lane's camera (-1: empty lane)
  std::vector<int> lane_acc;     // [tiles][64] >= 0: accumulator slot of the workgroup; < 0: ~(partial record) of a cold chunk
  std::vector<int> lane_seg;     // [tiles][64] first | last << 8 lane of the run that shares this lane's accumulator
  std::vector<int> bt_off;       // [grid * nb + 1] tiles of (workgroup, batch)
  std::vector<int> slot_rec;     // [lpl wg_cams.size()] partial record each workgroup slot is flushed to
  std::vector<int2> part_range;  // [n_cams] partial records of camera c (by camera index): [first, end)
  // for the bit-reproducible form of the kernel (e0_ck_det): ceil(log2(number of observations)) that are added into a
  // landmark slot ([lpl tiles][64], the lane order of V2::lmrec; 255: none -- a lane without a landmark, whose record is
  // undefined), and the ticket of every run total at its accumulator slot ([tiles][64], at the run's last lane): the order
  // is batches, then rounds of the tile walk over n_waves wavefronts, then the tiles of a round from the last (shortest)
  // to the first, then the lanes of a tile -- consistent with every wavefront's program order
  std::vector<uint8_t> lcnt_log2;
  std::vector<uint16_t> tick;
  int nb = 1;                    // batches per workgroup
  int li_mul = 3;                // the slot words hold li_mul x slot (CkShape)
  int ng = 1;                    // of which ng are in LDS at the same time (groups of wavefronts)
  int slots = 64;                // landmark slots of a batch (multiple of 64; the same for every workgroup)
  int stride = 0;                // fixed-stride shapes (step 2): the component stride of the LDS arrays the layout was cut for
  int n_part_rec = 0, max_acc = 0;  // max_acc: accumulator slots a workgroup has at most (the parent layout's, or the wide stride's cap)
  int64_t n_capped_obs = 0;      // observations of cameras with a slot in the parent layout but none here (wide stride)
  int64_t rows = 0, li_rows = 0;
  size_t n_uv = 0;               // entries of the row arrays (uv or uvp, src): (rows + CK_HMAX) x 64
  // statistics
  int64_t n_chunks = 0, n_cold_chunks = 0, n_obs = 0;
  int max_tiles_bt = 0;          // most tiles of a (workgroup, batch)
  double extra_lanes = 0;        // bank collisions left: extra lanes per (row, half), summed over the rows
};

// packs K.uv into K.uvp if every entry survives the round trip; false (and nothing changed) otherwise
inline bool ck_pack_uv(CkLayout& K, int n_threads) {
  const size_t n = K.uv.size();
  std::vector<int2> out(n);
  const int pieces = (int)std::min<size_t>(std::max<size_t>(n >> 16, 1), 4096);
  std::atomic<bool> ok{true};
  lpl_parallel(pieces, n_threads, [&](int pc) {
    const size_t i0 = n * (size_t)pc / pieces, i1 = n * (size_t)(pc + 1) / pieces;
    for (size_t i = i0; i < i1 && ok.load(std::memory_order_relaxed); ++i) {
      int32_t a = 0, b = 0;
      if (!ck_pack_one(K.uv[i].x, a) || !ck_pack_one(K.uv[i].y, b)) { ok.store(false); return; }
      out[i] = make_int2(a, b);
    }
  });
  if (!ok.load()) return false;
  K.uvp.swap(out);
  std::vector<double2>().swap(K.uv);
  K.packed = true;
  return true;
}

inline size_t ck_lds_bytes(int slots, int n_acc, int ng = 1) { return 16 + (size_t)ng * slots * 48 + (size_t)n_acc * 104 + 64; }  // = ck_lds_bytes_dev

// cost of one batch for chunk cap H: longest-first schedule of its tiles over n_waves wavefronts, every tile charged
// its height plus a fixed overhead (record gather, flush); also returns the tile count
inline int ck_schedule_cost(const std::vector<int>& counts, int H, int n_waves, int tile_overhead, int* n_tiles_out) {
  std::vector<int> len;
  for (int n : counts) {
    const int k = (n + H - 1) / H, base = n / k, rem = n % k;
    for (int q = 0; q < k; ++q) len.push_back(base + (q < rem ? 1 : 0));
  }
  std::sort(len.begin(), len.end(), std::greater<int>());
  const int n_tiles = ((int)len.size() + WAVE - 1) / WAVE;
  std::priority_queue<int, std::vector<int>, std::greater<int>> heap;
  for (int w = 0; w < n_waves; ++w) heap.push(0);
  int makespan = 0;
  for (int t = 0; t < n_tiles; ++t) {
    const int load = heap.top() + len[(size_t)t * WAVE] + tile_overhead;
    heap.pop();
    heap.push(load);
    makespan = std::max(makespan, load);
  }
  if (n_tiles_out) *n_tiles_out = n_tiles;
  return makespan;
}

// L: the lane-per-landmark layout (rows in the order the device will use); rank -> camera index through cam_of_rank
// n_waves, hmax: wavefronts per workgroup of the kernel instantiation that will run the layout, and the tallest tile it
// keeps in registers (CK_HMAX: none / whatever schedules best)
// ng: groups of wavefronts that work on different batches at the same time (the LDS holds ng batches; the batch count is a
// multiple of ng; n_waves = wavefronts of ONE group)
// shape: what the kernel that runs the layout keeps in LDS per landmark slot and how it addresses it.  Step 1 (e0_ck):
// 48 bytes, slot-major [slot][3] arrays (the slot words hold 3 x slot).  Step 2 (e0_ck_h): 64 bytes in component-major
// arrays with a compile-time stride of max_slots (the words hold the slot; the LDS need does not depend on the slot count).
struct CkShape {
  int slot_bytes = 48;
  int li_mul = 3;
  int max_slots = INT_MAX;
  bool need_uv = true;   // (step 2's operator does not read the image coordinates)
  int acc_bytes = 104;   // LDS bytes per accumulator slot (13 doubles)
  bool cold_q = false;   // chunks of cameras without a slot leave q per observation in the parent layout's cold view (CkLayout::cpos)
                         // instead of a 96-byte record of their own (step 1's e0_ck; the other kernels keep the records)
  int wide_slots = 0;    // fixed-stride shapes: a second stride, taken where it saves a landmark batch; the accumulators that no
                         // longer fit beside it are given up (the workgroup keeps the slots of its most observed cameras)
};
inline CkShape ck_shape_step1() { CkShape s; s.cold_q = true; return s; }
// e0_ck_det (povar_kernels_ck_det.hpp) keeps two more bytes per landmark slot (the sum's binary point) and per accumulator
// slot (the ticket counter)
inline CkShape ck_shape_det() { CkShape s; s.slot_bytes = 50; s.acc_bytes = 106; return s; }
inline CkShape ck_shape_step2_det();
// 1536 = CKH_STRIDE, 2048 = CKH_STRIDE_WIDE (povar_kernels_ck_joint.hpp); POVAR_CKH_STRIDE=1536|2048 forces one
inline CkShape ck_shape_step2() {
  CkShape s{64, 1, 1536, false};
  s.wide_slots = 2048;
  if (const char* e = std::getenv("POVAR_CKH_STRIDE")) {
    if (std::atoi(e) == 2048) s.max_slots = 2048;
    s.wide_slots = 0;
  }
  return s;
}
inline CkShape ck_shape_step2_det() { CkShape s{64, 1, 1536, false}; s.slot_bytes = 66; s.acc_bytes = 106; return s; }  // (e0_ck_h_det: one stride)
inline size_t ck_lds_bytes_shape(const CkShape& sh, int slots, int n_acc, int ng) {
  if (sh.max_slots != INT_MAX)  // = ckh_lds_bytes[_det]
    return 16 + (size_t)sh.slot_bytes * sh.max_slots + (size_t)n_acc * sh.acc_bytes + 64 + (sh.acc_bytes != 104 ? 16 : 0);
  return 16 + (size_t)ng * slots * sh.slot_bytes + (size_t)n_acc * sh.acc_bytes + 64 + (sh.acc_bytes != 104 ? 16 : 0);
}
// cold_q pays where cold observations are few: the lanes that serve them walk their (one- or two-row) tiles without the row
// prefetch and store 32 scattered bytes per observation.  venice-1778 Zipf(1) (3 % cold): 56.3 -> 55.0 us per term, `local` (1.5 %)
// 53.3 -> 52.8; Zipf(0.5) (18 %): 77.7 -> 88.0, uniform popularity (31 %): 80.0 -> 115.2 (profiles/r06_cold_q_ab.txt) -- so only
// up to CK_COLD_Q_MAX_PERCENT of the observations; beyond, a record per cold chunk.
constexpr int CK_COLD_Q_MAX_PERCENT = 8;
inline void build_ck(const LplLayout& L, int n_cams, int grid, const std::vector<int>& cam_of_rank, int n_waves,
                     CkLayout& K, bool place = true, int hmax = CK_HMAX, int ng = 1, const CkShape& shape_in = CkShape(),
                     bool pack_uv = true) {
  CkShape shape = shape_in;
  if (shape.cold_q) {
    int64_t n_obs_all = 0;
    for (int c : L.cw) n_obs_all += c != -1;
    if ((int64_t)L.cold_lm.size() * 100 > (int64_t)CK_COLD_Q_MAX_PERCENT * n_obs_all && !std::getenv("POVAR_CK_COLD_Q_ALWAYS")) shape.cold_q = false;
  }
  int n_threads = std::min(lpl_effective_cpus(), 128);
  if (const char* e = std::getenv("POVAR_LAYOUT_THREADS")) n_threads = std::max(1, std::atoi(e));
  hmax = std::min(CK_HMAX, std::max(1, hmax));
  // rows a tile is charged on top of its height in the schedule.  What a wavefront's SECOND tile of a batch really costs is
  // 6-11 thousand cycles of latency (metadata, record gather, first rows: profiles/r04_e0_ck_phase_stamps.txt), i.e. far
  // more than its rows: with 3 the schedule cut venice into 20 tiles per (workgroup, batch) -- four wavefronts walk two --,
  // from 8 on into at most 16 (15.96 k -> 16.5 k terms/s; 8 ... 32 within 0.5 %: profiles/r05_experiments.txt)
  int tile_overhead = 12;
  if (const char* e = std::getenv("POVAR_CK_TILE_COST")) tile_overhead = std::max(0, std::atoi(e));
  // ---- batches: the smallest count whose landmark slots fit next to the accumulators
  int max_tiles_w = 1, max_acc = 1;
  for (int w = 0; w < grid; ++w) {
    max_tiles_w = std::max(max_tiles_w, L.wg_tile_off[w + 1] - L.wg_tile_off[w]);
    max_acc = std::max(max_acc, L.wg_cam_off[w + 1] - L.wg_cam_off[w]);
  }
  ng = std::max(ng, 1);
  if (shape.max_slots != INT_MAX) {
    // fixed stride: the wide one where it saves a batch; then as many accumulators as fit beside the landmark slots
    auto batches = [&](int ms) { int n = 1; while (WAVE * ((max_tiles_w + n - 1) / n) > ms && n < max_tiles_w) ++n; return n; };
    if (shape.wide_slots > shape.max_slots && batches(shape.wide_slots) < batches(shape.max_slots)) shape.max_slots = shape.wide_slots;
    const size_t fixed = ck_lds_bytes_shape(shape, 0, 0, ng);
    max_acc = std::min(max_acc, (int)(((size_t)CK_LDS_BYTES - fixed) / (size_t)shape.acc_bytes));
    // (POVAR_CKH_ACC_CAP=<n>: stress knob of tools/forced_mode_suite.sh -- capped accumulators on problems with few cameras too)
    if (const char* e = std::getenv("POVAR_CKH_ACC_CAP"))
      if (shape.acc_bytes == 104) max_acc = std::min(max_acc, std::max(1, std::atoi(e)));
    K.stride = shape.max_slots;
  }
  int nb = ng;
  while ((ck_lds_bytes_shape(shape, WAVE * ((max_tiles_w + nb - 1) / nb), max_acc, ng) > (size_t)CK_LDS_BYTES ||
          WAVE * ((max_tiles_w + nb - 1) / nb) > shape.max_slots) && nb < max_tiles_w) nb += ng;
  if (const char* e = std::getenv("POVAR_CK_NB")) nb = std::max(nb, (std::atoi(e) + ng - 1) / ng * ng);
  K.nb = nb;
  K.li_mul = shape.li_mul;
  K.ng = ng;
  K.slots = WAVE * ((max_tiles_w + nb - 1) / nb);
  K.max_acc = max_acc;
  struct WgOut {
    std::vector<double2> uv;
    std::vector<uint32_t> li;
    std::vector<int> cpos;
    std::vector<int> src, lane_cam, lane_acc, lane_seg, bt_tiles;  // bt_tiles[b]: tiles of batch b
    std::vector<int4> tile;  // x, w: local row / li-row numbers
    std::vector<int> cold_rank;  // rank of every cold chunk, in the order their lanes say ~(index)
    int64_t chunks = 0, obs = 0, n_cold_q = 0, capped = 0;
    double extra = 0;
  };
  std::vector<WgOut> out(grid);
  std::vector<std::vector<int>> acc_rank_of(grid);
  std::vector<int> lcnt(L.tile.size() * WAVE, 0);  // (every workgroup writes its own tiles)
  lpl_parallel(grid, n_threads, [&](int w) {
    WgOut& o = out[w];
    o.bt_tiles.assign(nb, 0);
    const int t0 = L.wg_tile_off[w], t1 = L.wg_tile_off[w + 1];
    struct Ob { int key, li, src; };  // key: accumulator slot, or n_acc_w + rank for a camera without one
    // accumulator slots: the parent layout's, or -- where the LDS beside a wide stride holds fewer -- those of the cameras this
    // workgroup observes most (acc_of_slot: parent slot -> accumulator or -1; acc_rank: accumulator -> popularity rank)
    const int n_par_w = L.wg_cam_off[w + 1] - L.wg_cam_off[w];
    const int n_acc_w = std::min(n_par_w, max_acc);
    std::vector<int> acc_of_slot(n_par_w);
    std::vector<int>& acc_rank = acc_rank_of[w];
    acc_rank.resize(n_acc_w);
    if (n_acc_w == n_par_w) {
      for (int sl = 0; sl < n_par_w; ++sl) { acc_of_slot[sl] = sl; acc_rank[sl] = L.wg_cams[L.wg_cam_off[w] + sl]; }
    } else {
      std::vector<long> cnt(n_par_w, 0);
      for (int t = L.wg_tile_off[w]; t < L.wg_tile_off[w + 1]; ++t) {
        const int4 ti = L.tile[t];
        for (size_t idx = (size_t)ti.x * WAVE; idx < ((size_t)ti.x + ti.y) * WAVE; ++idx)
          if (L.cw[idx] >= 0) cnt[lpl_cw_slot(L.cw[idx])]++;
      }
      std::vector<int> by_cnt(n_par_w);
      for (int sl = 0; sl < n_par_w; ++sl) by_cnt[sl] = sl;
      std::stable_sort(by_cnt.begin(), by_cnt.end(), [&](int a, int c) { return cnt[a] > cnt[c]; });
      std::vector<char> keep(n_par_w, 0);
      for (int i = 0; i < n_acc_w; ++i) keep[by_cnt[i]] = 1;
      int a = 0;
      for (int sl = 0; sl < n_par_w; ++sl) {  // (kept slots stay in the parent's order: its deal over the LDS banks)
        acc_of_slot[sl] = keep[sl] ? a : -1;
        if (keep[sl]) acc_rank[a++] = L.wg_cams[L.wg_cam_off[w] + sl];
        else o.capped += cnt[sl];
      }
    }
    std::vector<Ob> obs;
    std::vector<int> counts, first_of;
    struct Chunk { int key, first, len; };
    std::vector<Chunk> chunks;
    std::vector<int> assign;
    std::vector<long> cost;
    for (int b = 0; b < nb; ++b) {
      obs.clear();
      for (int t = t0 + b; t < t1; t += nb) {
        const int4 ti = L.tile[t];
        const int slot0 = ((t - t0) / nb) * WAVE;
        for (int j = 0; j < ti.y; ++j)
          for (int lane = 0; lane < WAVE; ++lane) {
            const size_t idx = ((size_t)ti.x + j) * WAVE + lane;
            const int cw = L.cw[idx];
            if (cw == -1) continue;
            Ob ob;
            if (cw >= 0) {
              const int sl = lpl_cw_slot(cw), a = acc_of_slot[sl];
              ob.key = a >= 0 ? a : n_acc_w + L.wg_cams[L.wg_cam_off[w] + sl];
            } else {
              ob.key = n_acc_w + (-2 - cw);
            }
            ob.li = slot0 + (L.seg[(size_t)t * WAVE + lane] & 255);  // a landmark dealt over several lanes: its first lane's slot
            ob.src = (int)idx;
            obs.push_back(ob);
            lcnt[(size_t)t * WAVE + (L.seg[(size_t)t * WAVE + lane] & 255)]++;
          }
      }
      o.obs += (int64_t)obs.size();
      std::stable_sort(obs.begin(), obs.end(), [](const Ob& a, const Ob& c) { return a.key < c.key; });
      counts.clear();
      first_of.clear();
      for (size_t i = 0; i < obs.size();) {
        size_t j = i;
        while (j < obs.size() && obs[j].key == obs[i].key) ++j;
        counts.push_back((int)(j - i));
        first_of.push_back((int)i);
        i = j;
      }
      // chunk cap of this batch
      int H = hmax, best = INT_MAX;
      for (int h = std::min(2, hmax); h <= hmax; ++h) {
        const int c = ck_schedule_cost(counts, h, n_waves, tile_overhead, nullptr);
        if (c <= best) { best = c; H = h; }  // ties: the larger cap (fewer chunks)
      }
      chunks.clear();
      for (size_t g = 0; g < counts.size(); ++g) {
        const int n = counts[g], k = (n + H - 1) / H, base = n / k, rem = n % k;
        int at = first_of[g];
        for (int q = 0; q < k; ++q) {
          const int len = base + (q < rem ? 1 : 0);
          chunks.push_back(Chunk{obs[at].key, at, len});
          at += len;
        }
      }
      // inside a length: by popularity rank -- the lanes of a tile gather their camera records from the rank-ordered
      // image, and neighbouring ranks share cache lines (in slot order every lane of a gather hit a line of its own)
      auto rank_of = [&](int key) { return key >= n_acc_w ? key - n_acc_w : acc_rank[key]; };
      std::stable_sort(chunks.begin(), chunks.end(), [&](const Chunk& a, const Chunk& c) {
        if (a.len != c.len) return a.len > c.len;
        const int ra = rank_of(a.key), rc = rank_of(c.key);
        return ra != rc ? ra < rc : a.key < c.key;
      });
      o.chunks += (int64_t)chunks.size();
      const int n_tiles = ((int)chunks.size() + WAVE - 1) / WAVE;
      o.bt_tiles[b] = n_tiles;
      for (int tt = 0; tt < n_tiles; ++tt) {
        const int c0 = tt * WAVE, c1 = std::min((int)chunks.size(), c0 + WAVE);
        const int T = chunks[c0].len;
        int4 ti = make_int4((int)(o.uv.size() / WAVE), T, 0, (int)(o.li.size() / WAVE));
        const size_t r0 = o.uv.size(), q0 = o.li.size();
        o.uv.resize(r0 + (size_t)T * WAVE, make_double2(0, 0));
        if (shape.cold_q) o.cpos.resize(r0 + (size_t)T * WAVE, -1);
        o.src.resize(r0 + (size_t)T * WAVE, -1);
        o.li.resize(q0 + (size_t)((T + 1) / 2) * WAVE, CK_NONE | (CK_NONE << 16));
        const size_t l0 = o.lane_cam.size();
        o.lane_cam.resize(l0 + WAVE, -1);
        o.lane_acc.resize(l0 + WAVE, 0);
        o.lane_seg.resize(l0 + WAVE, 0);
        for (int lane = 0; lane < WAVE; ++lane) o.lane_seg[l0 + lane] = lane | (lane << 8);
        // bank occupancy of the tile: [row][half][32]
        uint16_t occ[CK_HMAX][2][32] = {};
        uint16_t mx[CK_HMAX][2];
        for (int j = 0; j < CK_HMAX; ++j) mx[j][0] = mx[j][1] = 1;
        for (int q = c0; q < c1; ++q) {
          const int lane = q - c0, half = lane >> 5;
          const Chunk& ck = chunks[q];
          const bool cold = ck.key >= n_acc_w;
          o.lane_cam[l0 + lane] = cold ? ck.key - n_acc_w : acc_rank[ck.key];
          if (cold) {
            if (shape.cold_q) {
              o.lane_acc[l0 + lane] = -1;  // no record: q goes to the cold view, observation by observation
              ++o.n_cold_q;
            } else {
              o.lane_acc[l0 + lane] = ~(int)o.cold_rank.size();
              o.cold_rank.push_back(ck.key - n_acc_w);
            }
            ti.z |= CK_FLAG_COLD;
          } else {
            o.lane_acc[l0 + lane] = ck.key;
            if (lane > 0 && chunks[q - 1].key == ck.key) ti.z |= CK_FLAG_DUP;
          }
          // rows of the chunk's observations: an assignment problem against the banks already taken in this half of
          // the tile (a lane with fewer observations than the tile has rows may leave any rows empty)
          const int h = ck.len;
          assign.resize(h);
          if (place && T > 1) {
            cost.assign((size_t)T * T, 0);
            for (int a = 0; a < h; ++a) {
              const int cls = obs[ck.first + a].li & 31;
              for (int j = 0; j < T; ++j) {
                const int v = occ[j][half][cls];
                cost[(size_t)a * T + j] = 1000L * std::max(0, v + 1 - (int)mx[j][half]) + v;
              }
            }
            std::vector<int> full;
            lpl_assign(T, cost, full);  // items h..T-1 are dummies (zero cost everywhere)
            for (int a = 0; a < h; ++a) assign[a] = full[a];
          } else {
            for (int a = 0; a < h; ++a) assign[a] = a;
          }
          for (int a = 0; a < h; ++a) {
            const Ob& ob = obs[ck.first + a];
            const int j = assign[a];
            const size_t idx = r0 + (size_t)j * WAVE + lane;
            const size_t s = (size_t)ob.src;
            o.uv[idx] = L.uv[s];
            if (shape.cold_q && cold) o.cpos[idx] = L.cpos[s];
            o.src[idx] = ob.src;
            uint32_t& word = o.li[q0 + (size_t)(j >> 1) * WAVE + lane];
            const uint32_t li3 = (uint32_t)shape.li_mul * (uint32_t)ob.li;  // step 1: the slot's first double in the [slot][3] LDS arrays
            word = (j & 1) ? ((word & 0xffffu) | (li3 << 16)) : ((word & 0xffff0000u) | li3);
            const uint16_t v = ++occ[j][half][ob.li & 31];
            mx[j][half] = std::max(mx[j][half], v);
          }
        }
        for (int j = 0; j < T; ++j) o.extra += (mx[j][0] - 1) + (mx[j][1] - 1);
        // runs of lanes with the same accumulator (resident chunks of one camera are adjacent: sorted by key)
        if (ti.z & CK_FLAG_DUP) {
          int lane = 0;
          const int nl = c1 - c0;
          while (lane < nl) {
            int e = lane;
            if (o.lane_acc[l0 + lane] >= 0)
              while (e + 1 < nl && o.lane_acc[l0 + e + 1] == o.lane_acc[l0 + lane]) ++e;
            for (int x = lane; x <= e; ++x) o.lane_seg[l0 + x] = lane | (e << 8);
            lane = e + 1;
          }
        }
        o.tile.push_back(ti);
      }
    }
  });
  // ---- global numbering
  K.bt_off.assign((size_t)grid * nb + 1, 0);
  std::vector<int64_t> row0_of(grid), li0_of(grid);
  std::vector<int> tile0_of(grid), cold0_of(grid);
  int n_cold_chunks = 0;
  int64_t n_cold_q = 0;
  for (int w = 0; w < grid; ++w) {
    const WgOut& o = out[w];
    row0_of[w] = K.rows;
    li0_of[w] = K.li_rows;
    tile0_of[w] = (int)K.tile.size();
    cold0_of[w] = n_cold_chunks;
    n_cold_chunks += (int)o.cold_rank.size();
    n_cold_q += o.n_cold_q;
    int t = (int)K.tile.size();
    for (int b = 0; b < nb; ++b) {
      K.bt_off[(size_t)w * nb + b] = t;
      t += o.bt_tiles[b];
      K.max_tiles_bt = std::max(K.max_tiles_bt, o.bt_tiles[b]);
    }
    for (int4 ti : o.tile) {
      ti.x += (int)K.rows;
      ti.w += (int)K.li_rows;
      K.tile.push_back(ti);
    }
    K.rows += (int64_t)(o.uv.size() / WAVE);
    K.li_rows += (int64_t)(o.li.size() / WAVE);
    K.n_chunks += o.chunks;
    K.n_obs += o.obs;
    K.n_capped_obs += o.capped;
    K.extra_lanes += o.extra;
  }
  K.bt_off[(size_t)grid * nb] = (int)K.tile.size();
  K.n_cold_chunks = n_cold_chunks + n_cold_q;  // (statistics: chunks of cameras without a slot, with or without a record)
  K.cold_q = shape.cold_q;
  auto clog2 = [](int n) { int e = 0; while ((1 << e) < n) ++e; return (uint8_t)(n == 0 ? 255 : e); };  // 255: nothing is added there
  K.lcnt_log2.resize(lcnt.size());
  for (size_t i = 0; i < lcnt.size(); ++i) K.lcnt_log2[i] = clog2(lcnt[i]);

  // ---- partial records, camera-major: a camera's workgroup slots first, then its cold chunks
  std::vector<int> n_rec(n_cams, 0);  // by rank
  for (int w = 0; w < grid; ++w)
    for (int r0 : acc_rank_of[w]) n_rec[r0]++;
  std::vector<int> n_slot_rec(n_rec);
  for (int w = 0; w < grid; ++w)
    for (int r0 : out[w].cold_rank) n_rec[r0]++;
  std::vector<int> first(n_cams + 1, 0);
  for (int r0 = 0; r0 < n_cams; ++r0) first[r0 + 1] = first[r0] + n_rec[r0];
  K.n_part_rec = first[n_cams];
  K.part_range.assign(n_cams, make_int2(0, 0));
  for (int r0 = 0; r0 < n_cams; ++r0) K.part_range[cam_of_rank[r0]] = make_int2(first[r0], first[r0 + 1]);
  K.slot_rec.assign(L.wg_cams.size(), 0);
  {
    std::vector<int> next(first.begin(), first.end() - 1);
    for (int w = 0; w < grid; ++w)  // (indexed by the parent's wg_cam_off + accumulator: a capped workgroup uses the first entries of its range)
      for (size_t a = 0; a < acc_rank_of[w].size(); ++a) K.slot_rec[(size_t)L.wg_cam_off[w] + a] = next[acc_rank_of[w][a]]++;
    // cold chunks: record numbers in workgroup order (deterministic whatever the thread count)
    std::vector<int> cold_rec((size_t)n_cold_chunks);
    for (int w = 0; w < grid; ++w)
      for (size_t i = 0; i < out[w].cold_rank.size(); ++i) cold_rec[(size_t)cold0_of[w] + i] = next[out[w].cold_rank[i]]++;
    // ---- concatenate the per-workgroup arrays
    // (+ CK_HMAX rows of padding: the register-resident tiles of e0_ck load CK_HMAX rows whatever the tile's height)
    K.uv.assign((size_t)(K.rows + CK_HMAX) * WAVE, make_double2(0, 0));
    K.src.assign((size_t)(K.rows + CK_HMAX) * WAVE, -1);
    if (shape.cold_q) K.cpos.assign((size_t)(K.rows + CK_HMAX) * WAVE, -1);
    K.li.assign((size_t)(K.li_rows + CK_HMAX) * WAVE, CK_NONE | (CK_NONE << 16));
    K.lane_cam.resize(K.tile.size() * WAVE);
    K.lane_acc.resize(K.tile.size() * WAVE);
    K.lane_seg.resize(K.tile.size() * WAVE);
    lpl_parallel(grid, n_threads, [&](int w) {
      const WgOut& o = out[w];
      std::copy(o.uv.begin(), o.uv.end(), K.uv.begin() + row0_of[w] * WAVE);
      std::copy(o.src.begin(), o.src.end(), K.src.begin() + row0_of[w] * WAVE);
      if (shape.cold_q) std::copy(o.cpos.begin(), o.cpos.end(), K.cpos.begin() + row0_of[w] * WAVE);
      std::copy(o.li.begin(), o.li.end(), K.li.begin() + li0_of[w] * WAVE);
      std::copy(o.lane_cam.begin(), o.lane_cam.end(), K.lane_cam.begin() + (size_t)tile0_of[w] * WAVE);
      std::copy(o.lane_seg.begin(), o.lane_seg.end(), K.lane_seg.begin() + (size_t)tile0_of[w] * WAVE);
      for (size_t i = 0; i < o.lane_acc.size(); ++i) {
        const int a = o.lane_acc[i];
        K.lane_acc[(size_t)tile0_of[w] * WAVE + i] = a >= 0 ? a : shape.cold_q ? -1 : ~cold_rec[(size_t)cold0_of[w] + (size_t)(~a)];
      }
    });
  }
  // ---- packed image points (e0_ck only: e0_ck_det keeps the 16-byte rows, step 2 reads none); POVAR_CK_PACK=0: never
  K.n_uv = K.uv.size();
  {
    const char* e = std::getenv("POVAR_CK_PACK");
    if (shape.need_uv && shape.slot_bytes == 48 && pack_uv && !(e && e[0] == '0')) ck_pack_uv(K, n_threads);
  }
  // ---- tickets of the run totals (e0_ck_det)
  K.tick.assign(K.tile.size() * WAVE, 0);
  lpl_parallel(grid, n_threads, [&](int w) {
    std::vector<int> next((size_t)std::max(1, L.wg_cam_off[w + 1] - L.wg_cam_off[w]), 0);
    for (int b = 0; b < nb; ++b) {
      const int tb0 = K.bt_off[(size_t)w * nb + b], tb1 = K.bt_off[(size_t)w * nb + b + 1];
      for (int q0 = tb0; q0 < tb1; q0 += n_waves)
        for (int t = std::min(q0 + n_waves, tb1) - 1; t >= q0; --t)
          for (int lane = 0; lane < WAVE; ++lane) {
            const size_t i = (size_t)t * WAVE + lane;
            if (K.lane_cam[i] >= 0 && K.lane_acc[i] >= 0 && lane == ((K.lane_seg[i] >> 8) & 255))
              K.tick[i] = (uint16_t)next[K.lane_acc[i]]++;
          }
    }
  });
}

}  // namespace povar
