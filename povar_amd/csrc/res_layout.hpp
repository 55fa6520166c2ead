// res_layout.hpp -- host-side construction of the layout of the RESIDENT power-series kernel series_res
// (povar_kernels_res.hpp: struct ResP).  Pure host C++, no device code.
//
// The per-term kernels (e0_lpl / e0_ck + cam_cold_sum_binv) are launched once per term and re-read everything that does
// not change between the terms of one solve_pOSE (linearization_power_varproj.hpp:191-237): observation rows, landmark
// records, tile metadata.  On contexts whose operands fit on the chip (ladybug-49, trafalgar-257, the landmark shards of
// venice-1778 from eight ranks on) that fixed cost is the whole term.  series_res is ONE launch per solve: a workgroup
// owns a set of landmarks for the whole series, the lanes keep their observation rows and their camera's P3 in registers,
// the landmarks (h~, u / g) and the accumulators of the cameras several lane runs share live in LDS, and per term only
// z = sigma x moves: the owners of the cameras publish it, every workgroup gathers the entries its lanes need.
//
//   * landmarks -> workgroups: contiguous ranges of a landmark ORDER.  Two orders are tried: the natural order (a real
//     reconstruction has locality: neighbouring landmarks share cameras) and the order by each landmark's RAREST camera
//     (a graph without locality, the SURVEY 8(d) generator: the hub cameras are everywhere anyway, the rare ones are
//     gathered into few workgroups).  The ranges are cut so that every workgroup has about the same number of lane
//     CHUNKS (what a workgroup's time and its register capacity are counted in), not of observations;
//   * chunk = at most H observations of ONE camera of the workgroup (as in ck_layout.hpp, with the rows in registers
//     instead of a row stream): a camera's run is cut into near-equal chunks, the chunks are sorted by (length, camera)
//     so that the lanes of a wavefront have the same number of rows and lanes of one camera are adjacent (one segmented
//     wavefront sum); a lane holds up to R chunks (rounds);
//   * every (workgroup, camera) pair has one partial record; a camera whose chunks are ONE run of adjacent lanes writes
//     it straight from registers, the others through an accumulator in LDS.  The records are camera-major, and every
//     camera has an OWNER workgroup (balanced by record count) that sums its records, applies B_c^-1 and publishes z_c.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "lpl_layout.hpp"

namespace povar {

constexpr int RES_HMAX = 8;          // rows of a chunk at most
constexpr int RES_MAX_WG = 256;      // workgroups (flag words swept by one wavefront: 4 per lane)
constexpr int RES_LDS_BYTES = 160 * 1024;
constexpr int RES_ACC_STRIDE = 13;   // doubles per accumulator slot in LDS (12 used; odd: 32 bank classes)

struct ResLayout {
  int W = 0, NW = 0, H = 0, R = 1;   // workgroups, wavefronts per workgroup, rows per chunk, chunks (rounds) per lane
  int LS = 1;                        // landmark slots per lane (slot s of a workgroup belongs to thread s % T)
  // per chunk position [W][R][T]
  std::vector<int> lane_cam;         // camera SLOT of the chunk in its workgroup (-1: none)
  std::vector<int> lane_seg;         // first | last << 8 lane (of the wavefront) of the run of lanes that share the camera
                                     // | 1 << 16: the camera's only run in the workgroup (plain store instead of an LDS add)
  // per row [W][R][H][T]
  std::vector<double2> uv;
  std::vector<int> lslot;            // 3 x landmark slot of the workgroup (-1: no observation)
  std::vector<int> oslot;            // wave-bin slot of the observation (robust weight: Dp::sw / V2::w through V2::of_slot)
  std::vector<int> wave_h;           // [W][R][NW] rows of the wavefront's chunks | needs a segmented sum << 8 | scan steps << 12
  // landmark slots of the workgroups
  std::vector<int> lm_off, lm_id;    // [W + 1], landmark of each slot
  // camera slots of the workgroups, most observed camera first.  Slot s of workgroup g IS partial record cam_off[g] + s:
  // the records are workgroup-major, a workgroup writes its own as one contiguous run
  std::vector<int> cam_off, cam_id;  // [W + 1], camera of each slot
  std::vector<int> cam_zi;           // row of the camera in the z table (popularity rank: the hub cameras share lines)
  // owners
  std::vector<int> own_off, own_cam; // [W + 1], cameras owned by each workgroup
  std::vector<int> own_zi;           // z-table row of each owned camera
  std::vector<int2> own_q;           // [first, end) positions of the camera's records in the workgroup's read list
  std::vector<int> oq_off, oq_rec;   // [W + 1]; the records a workgroup reads as an owner, camera after camera, each
                                     // camera's in workgroup order (the fixed order of its sum)
  int n_rec = 0;
  int max_lm = 0, max_cam = 0, max_own = 0, max_chunks = 0, max_oq = 0;
  int order = 0;                     // 0: natural landmark order, 1: by rarest camera
  size_t lds_bytes = 0;
  bool fits = false;
  const char* why = "";              // when it does not fit
};

// LDS of a workgroup: control words, h~ and u per landmark (G stays in the registers of the slot's lane), ONE region that
// is in turn the z of the workgroup's cameras, their accumulators, and the records it reads as an owner, and per owned
// camera B^-1, sigma, sum, term, E0 row, norms and five partial sums
constexpr int RES_LM_BYTES = 48;
constexpr int RES_OWN_DOUBLES = 144 + 12 + 12 + 12 + 12 + 2 + 5 * 12;
__host__ __device__ inline size_t res_region_doubles(int n_cam, int n_oq) {
  const size_t a = (size_t)n_cam * RES_ACC_STRIDE, b = (size_t)n_oq * 12;
  return a > b ? a : b;
}
inline size_t res_lds_bytes(int n_lm, int n_cam, int n_own, int n_oq) {
  return 64 + (size_t)n_lm * RES_LM_BYTES + res_region_doubles(n_cam, n_oq) * 8 + (size_t)n_own * (RES_OWN_DOUBLES * 8 + 16) +
         (size_t)(n_cam + n_oq) * 4 + 8;
}

// W workgroups of NW wavefronts whose lanes hold R chunks of at most H rows each; the smallest H <= hmax (a power of
// two, >= hmin) that fits is taken.  rank1[c] = 1 + popularity rank of camera c.
inline void build_res(int n_cams, int n_lms, const int32_t* lm_off, const int32_t* cam_idx, const double* obs,
                      const std::vector<int>& rank1, const std::vector<int>& slot_of_obs, int W, int NW, int R_, int hmin, int hmax,
                      int ls_max, ResLayout& R, int force_order = -1) {
  const int T = NW * WAVE;
  R = ResLayout();
  R.NW = NW;
  R.R = std::max(1, R_);
  const int cap_lanes = R.R * T;
  ls_max = std::max(1, std::min(ls_max, 2));
  W = std::max(1, std::min(W, RES_MAX_WG));
  hmax = std::max(1, std::min(hmax, RES_HMAX));
  hmin = std::max(1, std::min(hmin, hmax));
  if (n_cams > 65535) { R.why = "more than 65535 cameras"; return; }
  int n_threads = std::min(lpl_effective_cpus(), 64);
  if (const char* e = std::getenv("POVAR_LAYOUT_THREADS")) n_threads = std::max(1, std::atoi(e));
  // ---- landmark orders
  std::vector<int> ord[2];
  ord[0].resize(n_lms);
  std::iota(ord[0].begin(), ord[0].end(), 0);
  {
    std::vector<int64_t> key(n_lms);
    for (int l = 0; l < n_lms; ++l) {
      int r1 = 0, r2 = 0;  // the two largest popularity ranks (rarest cameras) of the landmark
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
        const int r = rank1[cam_idx[i]];
        if (r > r1) { r2 = r1; r1 = r; }
        else if (r > r2) r2 = r;
      }
      key[l] = ((int64_t)r1 << 32) | (uint32_t)r2;
    }
    ord[1] = ord[0];
    std::stable_sort(ord[1].begin(), ord[1].end(), [&](int a, int b) { return key[a] < key[b]; });
  }
  // Greedy cut of an order into ranges of at most `cap` chunks (chunk cap H), ls_max T landmarks and what the LDS holds next to
  // the owned cameras; returns the number of ranges (first[]: their starts) and the (range, camera) pairs
  const int own_guess = (n_cams + W - 1) / W + 2;
  const int oq_guess = (int)std::min<int64_t>((int64_t)n_cams * 2, lm_off[n_lms] / std::max(W, 1) / 5 + 64);  // (checked exactly at the end)
  auto cut = [&](const std::vector<int>& order, int H, int cap, std::vector<int>* first, int64_t* pairs_out) {
    std::vector<int> cnt(n_cams, 0), stamp(n_cams, -1);
    int groups = 0, chunks = 0, lms = 0, cams = 0;
    int64_t pairs = 0;
    if (first) first->assign(1, 0);
    for (int p = 0; p < n_lms; ++p) {
      const int l = order[p];
      // what the landmark adds to the open range
      int add_chunks = 0, add_cams = 0;
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
        const int c = cam_idx[i];
        if (stamp[c] != groups) { ++add_cams; ++add_chunks; }
        else if (cnt[c] % H == 0) ++add_chunks;
      }
      const bool over = lms > 0 && (chunks + add_chunks > cap || lms + 1 > ls_max * T ||
                                    res_lds_bytes(lms + 1, cams + add_cams, own_guess, oq_guess) > (size_t)RES_LDS_BYTES);
      if (over) {
        pairs += cams;
        ++groups;
        chunks = lms = cams = 0;
        if (first) first->push_back(p);
      }
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
        const int c = cam_idx[i];
        if (stamp[c] != groups) { stamp[c] = groups; cnt[c] = 0; ++cams; }
        if (cnt[c] % H == 0) ++chunks;
        ++cnt[c];
      }
      ++lms;
    }
    if (lms > 0) { pairs += cams; ++groups; }
    if (first) first->push_back(n_lms);
    if (pairs_out) *pairs_out = pairs;
    return groups;
  };
  // per (order, H): the smallest cap that needs at most W ranges
  struct Cand { int o, H, cap; int64_t pairs; bool ok; };
  std::vector<Cand> cands;
  for (int o = 0; o < 2; ++o)
    for (int H = hmin; H <= hmax; H *= 2) cands.push_back(Cand{o, H, 0, 0, false});
  lpl_parallel((int)cands.size(), n_threads, [&](int q) {
    Cand& cd = cands[q];
    if (force_order >= 0 && cd.o != force_order) return;
    if (cut(ord[cd.o], cd.H, cap_lanes, nullptr, nullptr) > W) return;
    int lo = 1, hi = cap_lanes;
    while (lo < hi) {
      const int mid = (lo + hi) / 2;
      if (cut(ord[cd.o], cd.H, mid, nullptr, nullptr) <= W) hi = mid;
      else lo = mid + 1;
    }
    cd.cap = lo;
    cut(ord[cd.o], cd.H, lo, nullptr, &cd.pairs);
    cd.ok = true;
  });
  int best = -1;
  for (int q = 0; q < (int)cands.size(); ++q) {
    if (!cands[q].ok) continue;
    if (best < 0 || cands[q].H < cands[best].H || (cands[q].H == cands[best].H && cands[q].pairs < cands[best].pairs)) best = q;
  }
  if (best < 0) { R.why = "the observations do not fit the lanes of the workgroups"; return; }
  const int H = cands[best].H;
  const std::vector<int>& order = ord[cands[best].o];
  std::vector<int> first;
  const int groups = cut(order, H, cands[best].cap, &first, nullptr);
  R.order = cands[best].o;
  R.H = H;
  R.W = W = groups;  // (no empty workgroups)
  // ---- per workgroup: landmark slots, the observations grouped by camera
  R.lm_off.assign(W + 1, 0);
  for (int g = 0; g < W; ++g) R.lm_off[g + 1] = R.lm_off[g] + (first[g + 1] - first[g]);
  R.lm_id.resize(n_lms);
  struct Ob { int cam, slot3, i; };
  struct Chunk { int len, cam, ci; size_t at; };
  std::vector<std::vector<Ob>> wg_obs(W);
  std::vector<std::vector<int>> wg_cams(W);
  std::vector<std::vector<Chunk>> wg_chunks(W);
  lpl_parallel(W, n_threads, [&](int g) {
    std::vector<Ob>& ob = wg_obs[g];
    for (int p = first[g]; p < first[g + 1]; ++p) {
      const int l = order[p], s = p - first[g];
      R.lm_id[R.lm_off[g] + s] = l;
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) ob.push_back(Ob{cam_idx[i], 3 * s, i});
    }
    std::stable_sort(ob.begin(), ob.end(), [&](const Ob& a, const Ob& b) { return rank1[a.cam] < rank1[b.cam]; });
    std::vector<Chunk>& ch = wg_chunks[g];
    for (size_t q = 0; q < ob.size();) {
      size_t e = q;
      while (e < ob.size() && ob[e].cam == ob[q].cam) ++e;
      const int ci = (int)wg_cams[g].size();
      wg_cams[g].push_back(ob[q].cam);
      const int n = (int)(e - q), k = (n + H - 1) / H, base = n / k, rem = n % k;
      for (int j = 0; j < k; ++j) {
        const int len = base + (j < rem ? 1 : 0);
        ch.push_back(Chunk{len, ob[q].cam, ci, q});
        q += len;
      }
    }
    // (length, camera): lanes of a wavefront have the same number of rows; chunks of one camera and length adjacent
    std::stable_sort(ch.begin(), ch.end(), [](const Chunk& a, const Chunk& b) { return a.len != b.len ? a.len > b.len : a.ci < b.ci; });
  });
  R.cam_off.assign(W + 1, 0);
  for (int g = 0; g < W; ++g) {
    R.cam_off[g + 1] = R.cam_off[g] + (int)wg_cams[g].size();
    R.max_lm = std::max(R.max_lm, R.lm_off[g + 1] - R.lm_off[g]);
    R.max_cam = std::max(R.max_cam, (int)wg_cams[g].size());
    R.max_chunks = std::max(R.max_chunks, (int)wg_chunks[g].size());
  }
  R.LS = std::max(1, (R.max_lm + T - 1) / T);
  if (R.LS > ls_max || R.max_chunks > cap_lanes) { R.why = "internal: the cut does not respect its caps"; return; }
  // ---- partial records (workgroup-major: record = the workgroup's camera slot) and owners
  std::vector<int> rec_cnt(n_cams, 0);
  for (int g = 0; g < W; ++g)
    for (int c : wg_cams[g]) rec_cnt[c]++;
  R.n_rec = R.cam_off[W];
  R.cam_id.resize(R.n_rec);
  R.cam_zi.resize(R.n_rec);
  for (int g = 0; g < W; ++g)
    for (size_t s = 0; s < wg_cams[g].size(); ++s) {
      R.cam_id[R.cam_off[g] + s] = wg_cams[g][s];
      R.cam_zi[R.cam_off[g] + s] = rank1[wg_cams[g][s]] - 1;
    }
  {
    // every camera (also one without observations in this shard: its x is still B^-1 times the exchanged sum) gets the
    // least loaded workgroup, most records first; load = records + a fixed cost per camera (B^-1, publication)
    std::vector<int> by(n_cams);
    std::iota(by.begin(), by.end(), 0);
    std::stable_sort(by.begin(), by.end(), [&](int a, int b) { return rec_cnt[a] > rec_cnt[b]; });
    std::vector<std::vector<int>> own(W);
    std::vector<std::pair<int64_t, int>> heap;
    for (int g = 0; g < W; ++g) heap.push_back({0, g});
    auto cmp = [](const std::pair<int64_t, int>& a, const std::pair<int64_t, int>& b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    std::vector<int> owner_of(n_cams, 0), own_idx(n_cams, 0);
    for (int c : by) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      auto& top = heap.back();
      owner_of[c] = top.second;
      own[top.second].push_back(c);
      top.first += rec_cnt[c] + 16;
      std::push_heap(heap.begin(), heap.end(), cmp);
    }
    R.own_off.assign(W + 1, 0);
    R.oq_off.assign(W + 1, 0);
    for (int g = 0; g < W; ++g) {
      R.own_off[g + 1] = R.own_off[g] + (int)own[g].size();
      R.max_own = std::max(R.max_own, (int)own[g].size());
      int q = 0;
      for (int c : own[g]) {
        own_idx[c] = (int)R.own_cam.size();
        R.own_cam.push_back(c);
        R.own_zi.push_back(rank1[c] - 1);
        R.own_q.push_back(make_int2(q, q + rec_cnt[c]));
        q += rec_cnt[c];
      }
      R.oq_off[g + 1] = R.oq_off[g] + q;
      R.max_oq = std::max(R.max_oq, q);
    }
    R.oq_rec.assign(R.n_rec, -1);
    std::vector<int> fill(n_cams, 0);
    for (int g = 0; g < W; ++g)  // (workgroup order: the order of a camera's sum)
      for (size_t s = 0; s < wg_cams[g].size(); ++s) {
        const int c = wg_cams[g][s], o = own_idx[c];
        R.oq_rec[(size_t)R.oq_off[owner_of[c]] + R.own_q[o].x + fill[c]++] = R.cam_off[g] + (int)s;
      }
  }
  // ---- lanes and rows
  const size_t n_pos = (size_t)W * R.R * T;
  R.lane_cam.assign(n_pos, -1);
  R.lane_seg.assign(n_pos, 0);
  R.uv.assign(n_pos * H, make_double2(0, 0));
  R.lslot.assign(n_pos * H, -1);
  R.oslot.assign(n_pos * H, -1);
  R.wave_h.assign((size_t)W * R.R * NW, 0);
  lpl_parallel(W, n_threads, [&](int g) {
    const std::vector<Ob>& ob = wg_obs[g];
    const std::vector<Chunk>& ch = wg_chunks[g];
    // chunk q -> (round, lane): round 0 takes the first T chunks in order, round 1 the rest in REVERSE lane order (the
    // wavefront with the longest chunks of round 0 gets the shortest of round 1)
    auto pos_of = [&](size_t q) {
      const int r = (int)(q / T), t = (int)(q % T);
      return std::make_pair(r, (r & 1) ? T - 1 - t : t);
    };
    std::vector<int> cam_of_pos((size_t)R.R * T, -1), ci_of_pos((size_t)R.R * T, -1);
    for (size_t q = 0; q < ch.size(); ++q) {
      const auto rt = pos_of(q);
      const size_t pos = (size_t)rt.first * T + rt.second;
      cam_of_pos[pos] = ch[q].cam;
      ci_of_pos[pos] = ch[q].ci;
      const size_t lane = ((size_t)g * R.R + rt.first) * T + rt.second;
      R.lane_cam[lane] = ch[q].ci;
      for (int j = 0; j < ch[q].len; ++j) {
        const Ob& o = ob[ch[q].at + j];
        const size_t row = (((size_t)g * R.R + rt.first) * H + j) * T + rt.second;
        R.uv[row] = make_double2(obs[2 * (size_t)o.i], obs[2 * (size_t)o.i + 1]);
        R.lslot[row] = o.slot3;
        R.oslot[row] = slot_of_obs[o.i];
      }
    }
    // runs of adjacent lanes with one camera inside a wavefront and round (one segmented wavefront sum, one LDS add)
    std::vector<int> runs(wg_cams[g].size(), 0), len_of_pos((size_t)R.R * T, 0);
    for (size_t q = 0; q < ch.size(); ++q) {
      const auto rt = pos_of(q);
      len_of_pos[(size_t)rt.first * T + rt.second] = ch[q].len;
    }
    for (int r = 0; r < R.R; ++r)
      for (int wv = 0; wv < NW; ++wv) {
        int h = 0, dup = 0, longest = 1;
        for (int l0 = 0; l0 < WAVE;) {
          const size_t p0 = (size_t)r * T + (size_t)wv * WAVE + l0;
          if (ci_of_pos[p0] < 0) { ++l0; continue; }
          int l1 = l0;
          while (l1 + 1 < WAVE && ci_of_pos[p0 + (l1 + 1 - l0)] == ci_of_pos[p0]) ++l1;
          for (int l = l0; l <= l1; ++l) {
            R.lane_seg[((size_t)g * R.R + r) * T + (size_t)wv * WAVE + l] = l0 | (l1 << 8);
            h = std::max(h, len_of_pos[(size_t)r * T + (size_t)wv * WAVE + l]);
          }
          runs[ci_of_pos[p0]]++;
          if (l1 > l0) dup = 1;
          longest = std::max(longest, l1 - l0 + 1);
          l0 = l1 + 1;
        }
        int steps = 0;
        while ((1 << steps) < std::min(longest, 16)) ++steps;
        R.wave_h[((size_t)g * R.R + r) * NW + wv] = h | (dup << 8) | (std::max(steps, 1) << 12);
      }
    for (size_t pos = 0; pos < (size_t)R.R * T; ++pos)
      if (ci_of_pos[pos] >= 0 && runs[ci_of_pos[pos]] == 1) R.lane_seg[(size_t)g * R.R * T + pos] |= 1 << 16;
  });
  R.lds_bytes = 0;
  for (int g = 0; g < W; ++g)
    R.lds_bytes = std::max(R.lds_bytes, res_lds_bytes(R.lm_off[g + 1] - R.lm_off[g], R.cam_off[g + 1] - R.cam_off[g],
                                                      R.own_off[g + 1] - R.own_off[g], R.oq_off[g + 1] - R.oq_off[g]));
  if (R.lds_bytes > (size_t)RES_LDS_BYTES) { R.why = "landmarks + accumulators + owned cameras exceed the LDS"; return; }
  R.fits = true;
}

}  // namespace povar
