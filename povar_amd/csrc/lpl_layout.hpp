// lpl_layout.hpp -- host-side construction of the lane-per-landmark layout of the per-term E0 kernel (e0_lpl,
// povar_kernels.hpp: struct V2).  Pure host C++, no device code.
//
// What is decided here, once per problem:
//   * which cameras each E0 workgroup keeps in LDS (record + accumulator slots).  The G most observed cameras are
//     resident everywhere; the next Tn ("tail") cameras are laid on an A x B grid of the workgroups (grid = A*B):
//     tail camera t has coordinates (t mod A, (t / A) mod B) and is resident in every workgroup of its column or
//     its row.  Any TWO tail cameras are then resident together in some workgroup, so a landmark with at most two
//     tail observations can be handed to a workgroup in which all its cameras are in LDS.  Only observations of
//     cameras beyond G + Tn, or third and later tail observations, stay "cold" (record gathered from L2, scatter
//     scalars through the cold camera-major view).  (G, Tn) maximise the covered observation count under the LDS
//     capacity.
//   * which workgroup processes which landmark: most constrained landmarks first, least loaded eligible workgroup.
//   * per workgroup: tiles of 64 lanes (landmarks sorted by rows per lane, long landmarks dealt over adjacent lanes),
//     the order of a landmark's observations over the rows (greedy LDS bank placement), the row stream arrays.
//   * where each workgroup's accumulator slots are flushed: partial records are laid out camera-major, so the
//     per-camera kernel reads one contiguous run per camera.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <queue>
#include <thread>
#include <vector>

#include "povar_kernels.hpp"

namespace povar {

struct LplLayout {
  std::vector<double2> uv;
  std::vector<int> cw, cpos, seg, lm_pos, of_slot, lm_of;  // lm_of: [n_tiles][64] landmark of each lane (-1: none)
  std::vector<int4> tile;
  std::vector<int> wg_tile_off;  // [grid + 1]
  std::vector<int> wg_cam_off;   // [grid + 1] into wg_cams / wg_slot_rec
  std::vector<int> wg_cams;      // popularity rank (0-based) = index in the record image of the camera in each slot
  std::vector<int> wg_slot_rec;  // partial record each slot is flushed to
  std::vector<int2> part_range;  // [n_cams] partial records of camera c: [first, end)
  std::vector<int> cold_lm;      // [n_cold] landmark of each cold observation, camera-major
  std::vector<int2> cold_range;  // [n_cams] run of each camera in the cold view
  int64_t rows = 0;
  int n_part_rec = 0, max_slots = 0, n_global = 0, n_tail = 0, grid_a = 1, grid_b = 1;
};

// Minimum-cost assignment of h items to h positions (Hungarian algorithm, O(h^3); h is the number of resident
// observations of one landmark: 2..8 for almost all of them).  assign[item] = position.
inline void lpl_assign(int h, const std::vector<long>& cost, std::vector<int>& assign) {
  const long INF = 1L << 60;
  std::vector<long> u(h + 1, 0), v(h + 1, 0), minv(h + 1);
  std::vector<int> p(h + 1, 0), way(h + 1, 0);
  std::vector<char> used(h + 1);
  for (int i = 1; i <= h; ++i) {
    p[0] = i;
    int j0 = 0;
    std::fill(minv.begin(), minv.end(), INF);
    std::fill(used.begin(), used.end(), 0);
    do {
      used[j0] = 1;
      const int i0 = p[j0];
      long delta = INF;
      int j1 = 0;
      for (int j = 1; j <= h; ++j)
        if (!used[j]) {
          const long cur = cost[(size_t)(i0 - 1) * h + (j - 1)] - u[i0] - v[j];
          if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
          if (minv[j] < delta) { delta = minv[j]; j1 = j; }
        }
      for (int j = 0; j <= h; ++j)
        if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
        else minv[j] -= delta;
      j0 = j1;
    } while (p[j0] != 0);
    do {
      const int j1 = way[j0];
      p[j0] = p[j1];
      j0 = j1;
    } while (j0);
  }
  assign.assign(h, 0);
  for (int j = 1; j <= h; ++j) assign[p[j] - 1] = j - 1;
}

// rank1[c]: 1-based popularity rank of camera c; cnt_sorted[r]: observation count of the camera with rank r (0-based)
inline void build_lpl(int n_cams, int n_lms, const int32_t* lm_off, const int32_t* cam_idx, const double* obs,
                      const std::vector<int>& rank1, const std::vector<int>& slot_of_obs, size_t n_slots, int grid,
                      int n_acc, LplLayout& L) {
  // Rows per tile are capped by dealing longer landmarks over several lanes (K0 rows: at most 2 K0 row steps per
  // tile, the unit of load balance).  A wavefront walks its tile's rows one after the other, so on a problem too small
  // to give every wavefront a tile the cap sets the latency of the launch: 2 rows there (ladybug-49 76 k -> 106 k
  // terms/s, trafalgar-257 61 k -> 67 k), 8 otherwise (shorter tiles cost venice shards 2-3 us: more lanes, more
  // landmark records).
  grid = std::max(grid, 1);
  int n_with_obs = 0;
  for (int l = 0; l < n_lms; ++l) n_with_obs += lm_off[l + 1] > lm_off[l];
  int K0 = n_with_obs / WAVE < grid * 16 / 2 ? 2 : 8;
  if (const char* e = std::getenv("POVAR_LPL_K0")) K0 = std::max(2, std::atoi(e));
  // ---- grid factorisation and the (G, Tn) choice
  int B = 1;
  for (int b = 1; (int64_t)b * b <= grid; ++b)
    if (grid % b == 0) B = b;
  const int A = grid / B;
  std::vector<int64_t> S(n_cams + 1, 0);  // prefix sums of the observation counts in popularity order
  {
    std::vector<int64_t> cnt(n_cams, 0);
    for (int64_t i = 0; i < lm_off[n_lms]; ++i) cnt[rank1[cam_idx[i]] - 1]++;
    for (int r = 0; r < n_cams; ++r) S[r + 1] = S[r] + cnt[r];
  }
  int G = std::min(n_cams, n_acc), Tn = 0;
  if (n_cams > n_acc && std::getenv("POVAR_LPL_NOGRID") == nullptr) {
    double best = -1;
    for (int g = 0; g <= n_acc; g += 4) {
      const int cap = n_acc - g;
      // slots a workgroup needs: its column (<= Tn/A + 1 cameras) and its row (<= Tn/B + A: blocks of A ranks)
      int tn = (int)std::max(0.0, (cap - A - 2) / (1.0 / A + 1.0 / B));
      tn = std::min(tn, n_cams - g);
      // a landmark's first two tail observations are always resident; count later ones as half covered
      const double score = (double)S[g] + 0.85 * (double)(S[g + tn] - S[g]);
      if (score > best) { best = score; G = g; Tn = tn; }
    }
  }
  L.n_global = G;
  L.n_tail = Tn;
  L.grid_a = A;
  L.grid_b = B;
  auto tail_xy = [&](int r0, int& x, int& y) {  // r0: 0-based rank; false if not a grid camera
    const int t = r0 - G;
    if (t < 0 || t >= Tn) return false;
    x = t % A;
    y = (t / A) % B;
    return true;
  };
  auto resident = [&](int w, int r0) {
    if (r0 < G) return true;
    int x, y;
    if (!tail_xy(r0, x, y)) return false;
    return x == w % A || y == w / A;
  };
  // ---- landmark -> workgroup
  std::vector<int> wg_of(n_lms, -1);
  std::vector<int64_t> load(grid, 0);
  {
    std::vector<int> n_tail_of(n_lms, 0), order;
    for (int l = 0; l < n_lms; ++l) {
      if (lm_off[l + 1] == lm_off[l]) continue;
      int x, y;
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) n_tail_of[l] += tail_xy(rank1[cam_idx[i]] - 1, x, y);
      order.push_back(l);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_tail_of[a] > n_tail_of[b]; });
    std::vector<int> xs, ys;
    size_t pos = 0;
    for (; pos < order.size() && n_tail_of[order[pos]] > 0; ++pos) {
      const int l = order[pos], k = lm_off[l + 1] - lm_off[l];
      xs.clear();
      ys.clear();
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
        int x, y;
        if (tail_xy(rank1[cam_idx[i]] - 1, x, y)) { xs.push_back(x); ys.push_back(y); }
      }
      const int nt = (int)xs.size(), nc = std::min(nt, 8);
      int best_w = -1, best_cov = -1;
      auto consider = [&](int a, int b) {
        int cov = 0;
        for (int m = 0; m < nt; ++m) cov += xs[m] == a || ys[m] == b;
        const int w = a + A * b;
        if (cov > best_cov || (cov == best_cov && load[w] < load[best_w])) { best_cov = cov; best_w = w; }
      };
      if (nt == 1) {  // any workgroup of the column or the row
        for (int b = 0; b < B; ++b) consider(xs[0], b);
        for (int a = 0; a < A; ++a) consider(a, ys[0]);
      } else {
        for (int i = 0; i < nc; ++i)
          for (int j = 0; j < nc; ++j) consider(xs[i], ys[j]);
      }
      wg_of[l] = best_w;
      load[best_w] += k;
    }
    // landmarks without tail observations fill the workgroups up, longest first into the least loaded
    std::priority_queue<std::pair<int64_t, int>, std::vector<std::pair<int64_t, int>>, std::greater<>> heap;
    for (int w = 0; w < grid; ++w) heap.push({load[w], w});
    std::vector<int> rest(order.begin() + pos, order.end());
    std::stable_sort(rest.begin(), rest.end(),
                     [&](int a, int b) { return lm_off[a + 1] - lm_off[a] > lm_off[b + 1] - lm_off[b]; });
    for (int l : rest) {
      auto [ld, w] = heap.top();
      heap.pop();
      wg_of[l] = w;
      load[w] = ld + (lm_off[l + 1] - lm_off[l]);
      heap.push({load[w], w});
    }
  }
  // ---- cold view: observations whose camera is not resident in their landmark's workgroup, camera-major
  const int64_t n_obs = lm_off[n_lms];
  std::vector<int> cold_pos_of_obs(n_obs, -1);
  {
    std::vector<int> ccnt(n_cams + 1, 0);
    for (int l = 0; l < n_lms; ++l)
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i)
        if (!resident(wg_of[l], rank1[cam_idx[i]] - 1)) ccnt[cam_idx[i] + 1]++;
    for (int c = 0; c < n_cams; ++c) ccnt[c + 1] += ccnt[c];
    L.cold_range.resize(n_cams);
    for (int c = 0; c < n_cams; ++c) L.cold_range[c] = make_int2(ccnt[c], ccnt[c + 1]);
    L.cold_lm.resize(ccnt[n_cams]);
    std::vector<int> fill(ccnt.begin(), ccnt.end() - 1);
    for (int l = 0; l < n_lms; ++l)
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i)
        if (!resident(wg_of[l], rank1[cam_idx[i]] - 1)) {
          const int p = fill[cam_idx[i]]++;
          cold_pos_of_obs[i] = p;
          L.cold_lm[p] = l;
        }
  }
  // ---- per workgroup: camera slots (global cameras in rank order, then the tail cameras its landmarks use), tiles
  std::vector<std::vector<int>> lms_of(grid);
  for (int l = 0; l < n_lms; ++l)
    if (wg_of[l] >= 0) lms_of[wg_of[l]].push_back(l);
  L.wg_cam_off.assign(grid + 1, 0);
  L.wg_tile_off.assign(grid + 1, 0);
  L.lm_pos.assign(n_lms, -1);
  L.of_slot.assign(n_slots, -1);
  std::vector<std::vector<int>> holders(n_cams);  // rank -> workgroups with a slot for it (for the partial records)
  std::vector<int> parts_of(n_lms, 0), psize_of(n_lms, 0), cold_of(n_lms, 0);
  std::vector<std::vector<int>> order_of(grid);
  {
    std::vector<char> mark(n_cams, 0);
    for (int w = 0; w < grid; ++w) {
      const int slot0 = (int)L.wg_cams.size();
      // slots: the replicated hub cameras always (fixed slots 0..hubs-1), then only the resident cameras this
      // workgroup's landmarks actually observe, in rank order (a small shard touches far fewer than G + its grid
      // cameras: fewer records to stage, fewer partial records to flush and to sum)
      const int hubs_w = lpl_hubs(G);
      std::vector<int> used;
      for (int l : lms_of[w])
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
          const int r0 = rank1[cam_idx[i]] - 1;
          if (r0 >= hubs_w && !mark[r0] && resident(w, r0)) {
            mark[r0] = 1;
            used.push_back(r0);
          }
        }
      std::sort(used.begin(), used.end());
      for (int r0 = 0; r0 < hubs_w; ++r0) L.wg_cams.push_back(r0);
      // Which slot a camera gets decides its LDS bank (accumulators: (slot + 3 hubs) mod 32, records: slot mod 16).
      // In rank order the popular cameras pile up on the low banks and force collisions no row placement can avoid
      // (a bank hit by more observations than the tile has rows must repeat inside a row).  So the cameras are dealt
      // to the banks heaviest first, each to the least loaded bank that still has a free slot (weights = this
      // workgroup's observations per camera; the hubs' four replicas are pre-loaded with a quarter each).
      {
        const int n_rest = (int)used.size(), n_w_ = hubs_w + n_rest;
        std::vector<long> wcnt(n_cams, 0);
        for (int l : lms_of[w])
          for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) wcnt[rank1[cam_idx[i]] - 1]++;
        std::vector<double> load(32, 0.0);
        for (int r0 = 0; r0 < hubs_w; ++r0)
          for (int q = 0; q < 4; ++q) load[(4 * r0 + q) & 31] += 0.25 * (double)wcnt[r0];
        std::vector<std::vector<int>> free_slots(32);  // by accumulator bank, ascending
        for (int sl = n_w_ - 1; sl >= hubs_w; --sl) free_slots[(sl + 3 * hubs_w) & 31].push_back(sl);
        std::vector<int> by_weight(used);
        std::stable_sort(by_weight.begin(), by_weight.end(), [&](int a, int b) { return wcnt[a] > wcnt[b]; });
        std::vector<int> cam_of_slot(n_w_, -1);
        for (int r0 : by_weight) {
          int best = -1;
          for (int b = 0; b < 32; ++b)
            if (!free_slots[b].empty() && (best < 0 || load[b] < load[best])) best = b;
          const int sl = free_slots[best].back();
          free_slots[best].pop_back();
          cam_of_slot[sl] = r0;
          load[best] += (double)wcnt[r0];
        }
        for (int sl = hubs_w; sl < n_w_; ++sl) L.wg_cams.push_back(cam_of_slot[sl]);
        for (int r0 : used) mark[r0] = 0;
      }
      const int n_w = (int)L.wg_cams.size() - slot0;
      L.max_slots = std::max(L.max_slots, n_w);
      for (int s = 0; s < n_w; ++s) holders[L.wg_cams[slot0 + s]].push_back(w);
      L.wg_cam_off[w + 1] = (int)L.wg_cams.size();
      // tiles: lanes sorted by (rows per lane, cold rows per lane), longest first
      std::vector<int>& order = order_of[w];
      for (int l : lms_of[w]) {
        const int k = lm_off[l + 1] - lm_off[l];
        int cold = 0;
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) cold += cold_pos_of_obs[i] >= 0;
        parts_of[l] = k <= K0 ? 1 : std::min(WAVE, (k + K0 - 1) / K0);
        psize_of[l] = (k + parts_of[l] - 1) / parts_of[l];
        cold_of[l] = (cold + parts_of[l] - 1) / parts_of[l];
        order.push_back(l);
      }
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return psize_of[a] != psize_of[b] ? psize_of[a] > psize_of[b] : cold_of[a] > cold_of[b];
      });
      const int tile0 = (int)L.tile.size();
      int tile = tile0, fill = 0;
      for (int l : order) {
        if (fill + parts_of[l] > WAVE) { ++tile; fill = 0; }
        L.lm_pos[l] = (tile * WAVE + fill) | ((parts_of[l] - 1) << 26);
        fill += parts_of[l];
      }
      const int n_tiles_w = order.empty() ? 0 : tile - tile0 + 1;
      L.tile.resize(tile0 + n_tiles_w, make_int4(0, 0, 1 << 30, 0));
      L.seg.resize((size_t)(tile0 + n_tiles_w) * WAVE);
      L.lm_of.resize((size_t)(tile0 + n_tiles_w) * WAVE, -1);
      for (size_t i = (size_t)tile0 * WAVE; i < L.seg.size(); ++i) L.seg[i] = (int)(i & 63) | ((int)(i & 63) << 8);
      std::vector<int> lanes_used(n_tiles_w, 0);
      for (int l : order) {
        const int pos = L.lm_pos[l] & ((1 << 26) - 1), t = pos >> 6, lane0 = pos & 63, P = parts_of[l];
        int hot = 0;
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) hot += cold_pos_of_obs[i] < 0;
        int4& ti = L.tile[t];
        ti.y = std::max(ti.y, psize_of[l]);
        ti.z = std::min(ti.z, hot / P);  // leading rows in which every lane of the landmark has a resident camera
        if (P > 1) ti.w |= 1;
        for (int q = 0; q < P; ++q) {
          L.seg[(size_t)t * WAVE + lane0 + q] = lane0 | ((lane0 + P - 1) << 8);
          L.lm_of[(size_t)t * WAVE + lane0 + q] = l;
        }
        lanes_used[t - tile0] += P;
      }
      for (int t = tile0; t < tile0 + n_tiles_w; ++t) {
        if (lanes_used[t - tile0] < WAVE) L.tile[t].z = 0;  // unused lanes: no branch-free rows
        // at least two rows = four row steps per tile: the prefetch cursor (three rows ahead) then never needs a
        // tile beyond the one the consumer has already taken
        L.tile[t].y = std::max(L.tile[t].y, 2);
        L.tile[t].x = (int)L.rows;
        L.rows += L.tile[t].y;
      }
      L.wg_tile_off[w + 1] = (int)L.tile.size();
    }
  }
  L.uv.assign((size_t)L.rows * WAVE, make_double2(0, 0));
  L.cw.assign((size_t)L.rows * WAVE, -1);
  L.cpos.assign((size_t)L.rows * WAVE, -1);
  // ---- row stream with LDS bank placement, workgroups in parallel (disjoint row ranges).
  // ds_add_f64 takes 8 LDS cycles per wavefront when the 32 lanes of each half hit 32 different bank pairs and 8
  // more for every additional lane on a bank (tools/micro/lds_atomic_rates.hip); a ds_read_b128 takes one more
  // cycle per lane group for every additional record on a bank quad.  With the observations in their natural order
  // row 0 would hold every landmark's lowest-index camera.  So the assignment of a landmark's resident
  // observations to the rows of its lanes is an assignment problem against the banks already taken in its tile
  // (accumulator bank = lpl_acc_slot(slot, lane) mod 32 per row half, record quad = slot mod 16 per read group):
  // solved exactly per landmark (Hungarian), landmark after landmark, then once more with the whole tile known.
  auto read_group = [](int lane) {
    const int l = lane & 31;
    const int g = (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1;
    return g + 2 * (lane >> 5);
  };
  const int hubs = lpl_hubs(G);
  int n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (const char* e = std::getenv("POVAR_LAYOUT_THREADS")) n_threads = std::max(1, std::atoi(e));
  const bool no_place = std::getenv("POVAR_LPL_NOPLACE") != nullptr;  // measurement knob: natural order
  std::atomic<int> next_wg{0};
  auto worker = [&]() {
    std::vector<int> slot_of_rank(n_cams, -1);
    std::vector<int> hot_idx, cold_idx, assign, assign2, lane_of;
    std::vector<long> cost;
    std::vector<uint16_t> occA, occR;
    std::vector<std::vector<int>> placed;  // per landmark of the current tile: its hot observations in position order
    for (;;) {
      const int w = next_wg.fetch_add(1);
      if (w >= grid) break;
      const int slot0 = L.wg_cam_off[w], n_w = L.wg_cam_off[w + 1] - slot0;
      for (int s = 0; s < n_w; ++s) slot_of_rank[L.wg_cams[slot0 + s]] = s;
      const std::vector<int>& order = order_of[w];
      size_t o0 = 0;
      while (o0 < order.size()) {
        // landmarks of one tile
        const int t = (L.lm_pos[order[o0]] & ((1 << 26) - 1)) >> 6;
        size_t o1 = o0;
        while (o1 < order.size() && ((L.lm_pos[order[o1]] & ((1 << 26) - 1)) >> 6) == t) ++o1;
        const int R = L.tile[t].y;
        occA.assign((size_t)R * 64, 0);
        occR.assign((size_t)R * 64, 0);
        placed.assign(o1 - o0, {});
        auto bankA = [&](int s, int lane, int j) { return (size_t)j * 64 + (lane >> 5) * 32 + (lpl_acc_slot(s, lane, hubs) & 31); };
        auto bankR = [&](int s, int lane, int j) { return (size_t)j * 64 + read_group(lane) * 16 + (s & 15); };
        // single-lane landmarks may also take any of the tile's single lanes: which half, which ds_read_b128 lane
        // group and which hub replica (lane & 3) a landmark sits in decide its collisions, so every class of free
        // lanes is tried (16 classes) and the cheapest kept
        auto lane_class = [&](int lane) { return (lane >> 5) * 8 + (read_group(lane) & 1) * 4 + (lane & 3); };
        lane_of.assign(o1 - o0, 0);
        unsigned long long free_mask = 0;
        for (size_t o = o0; o < o1; ++o) {
          lane_of[o - o0] = L.lm_pos[order[o]] & 63;
          if (parts_of[order[o]] == 1) free_mask |= 1ull << lane_of[o - o0];
        }
        auto place_cost = [&](const std::vector<int>& cur, int lane0, int P, std::vector<int>& out_assign) -> long {
          const int h = (int)cur.size();
          if (h == 0) return 0;
          cost.assign((size_t)h * h, 0);
          for (int a = 0; a < h; ++a) {
            const int s = slot_of_rank[rank1[cam_idx[cur[a]]] - 1];
            for (int pos = 0; pos < h; ++pos) {
              const int lane = lane0 + pos % P, j = pos / P;
              long c = 96L * occA[bankA(s, lane, j)];
              if (s >= hubs) c += 16L * occR[bankR(s, lane, j)];
              cost[(size_t)a * h + pos] = c;
            }
          }
          lpl_assign(h, cost, out_assign);
          long tot = 0;
          for (int a = 0; a < h; ++a) tot += cost[(size_t)a * h + out_assign[a]];
          return tot;
        };
        for (int pass = 0; pass < 2; ++pass)
          for (size_t o = o0; o < o1; ++o) {
            const int l = order[o], P = parts_of[l];
            int lane0 = lane_of[o - o0];
            std::vector<int>& cur = placed[o - o0];
            if (pass == 0) {
              for (int i = lm_off[l]; i < lm_off[l + 1]; ++i)
                if (cold_pos_of_obs[i] < 0) cur.push_back(i);
            } else {  // take this landmark's banks out again, then place it against everything else
              for (size_t n = 0; n < cur.size(); ++n) {
                const int s = slot_of_rank[rank1[cam_idx[cur[n]]] - 1], lane = lane0 + (int)(n % P), j = (int)(n / P);
                occA[bankA(s, lane, j)]--;
                if (s >= hubs) occR[bankR(s, lane, j)]--;
              }
            }
            const int h = (int)cur.size();
            if (h <= 64 && !no_place) {
              if (pass == 0 && P == 1) {
                long best = -1;
                int best_lane = -1;
                unsigned seen = 0;
                for (unsigned long long m = free_mask; m; m &= m - 1) {
                  const int lane = __builtin_ctzll(m), cls = lane_class(lane);
                  if (seen & (1u << cls)) continue;
                  seen |= 1u << cls;
                  const long c = place_cost(cur, lane, 1, assign2);
                  if (best < 0 || c < best) { best = c; best_lane = lane; assign = assign2; }
                  if (best == 0) break;
                }
                lane0 = best_lane;
                free_mask &= ~(1ull << lane0);
                lane_of[o - o0] = lane0;
              } else {
                place_cost(cur, lane0, P, assign);
              }
              if (h > 1) {
                hot_idx.assign(h, 0);
                for (int a = 0; a < h; ++a) hot_idx[assign[a]] = cur[a];
                cur = hot_idx;
              }
            } else if (pass == 0 && P == 1) {
              free_mask &= ~(1ull << lane0);
            }
            for (int n = 0; n < h; ++n) {
              const int s = slot_of_rank[rank1[cam_idx[cur[n]]] - 1], lane = lane0 + n % P, j = n / P;
              occA[bankA(s, lane, j)]++;
              if (s >= hubs) occR[bankR(s, lane, j)]++;
            }
          }
        for (size_t o = o0; o < o1; ++o) {  // the lanes the single-lane landmarks ended up in
          const int l = order[o];
          if (parts_of[l] != 1) continue;
          L.lm_pos[l] = t * WAVE + lane_of[o - o0];
          L.lm_of[(size_t)t * WAVE + lane_of[o - o0]] = l;
        }
        // write the tile's rows
        for (size_t o = o0; o < o1; ++o) {
          const int l = order[o], lane0 = L.lm_pos[l] & 63, P = parts_of[l];
          const std::vector<int>& cur = placed[o - o0];
          const int h = (int)cur.size();
          cold_idx.clear();
          for (int i = lm_off[l]; i < lm_off[l + 1]; ++i)
            if (cold_pos_of_obs[i] >= 0) cold_idx.push_back(i);
          for (int n = 0; n < h + (int)cold_idx.size(); ++n) {
            const int i = n < h ? cur[n] : cold_idx[n - h];
            const int q = n % P, j = n / P, r0 = rank1[cam_idx[i]] - 1, lane = lane0 + q;
            const size_t idx = ((size_t)L.tile[t].x + j) * WAVE + lane;
            L.uv[idx] = make_double2(obs[2 * (size_t)i], obs[2 * (size_t)i + 1]);
            L.of_slot[slot_of_obs[i]] = (int)idx;
            if (n < h) {
              L.cw[idx] = slot_of_rank[r0];
            } else {
              L.cw[idx] = -2 - r0;  // cold: the record is gathered from the rank-ordered image
              L.cpos[idx] = cold_pos_of_obs[i];
            }
          }
        }
        o0 = o1;
      }
      for (int s = 0; s < n_w; ++s) slot_of_rank[L.wg_cams[slot0 + s]] = -1;
    }
  };
  {
    std::vector<std::thread> pool;
    for (int i = 1; i < n_threads; ++i) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  // ---- partial records, camera-major: camera c's slots in the workgroups that hold it form one contiguous run, so
  // the per-camera kernel streams them.  (Workgroup-major records -- one contiguous 53 KB flush per workgroup, the
  // per-camera kernel gathering through an index list -- were measured too: the flush is bound by the 13.5 MB it
  // writes, not by its access pattern, and the gather cost the per-camera kernel 1.6 us.)
  L.part_range.assign(n_cams, make_int2(0, 0));
  L.wg_slot_rec.assign(L.wg_cams.size(), 0);
  {
    std::vector<int> cam_of_rank(n_cams);
    for (int c = 0; c < n_cams; ++c) cam_of_rank[rank1[c] - 1] = c;
    std::vector<int> next(n_cams, 0);  // by rank
    int rec = 0;
    for (int r0 = 0; r0 < n_cams; ++r0) {
      next[r0] = rec;
      L.part_range[cam_of_rank[r0]] = make_int2(rec, rec + (int)holders[r0].size());
      rec += (int)holders[r0].size();
    }
    L.n_part_rec = rec;
    for (int w = 0; w < grid; ++w)
      for (int s = L.wg_cam_off[w]; s < L.wg_cam_off[w + 1]; ++s) L.wg_slot_rec[s] = next[L.wg_cams[s]]++;
  }
}

}  // namespace povar
