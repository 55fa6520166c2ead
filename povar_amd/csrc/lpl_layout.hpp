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
#include <chrono>
#include <cstdio>
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
  std::vector<int> cold_src;     // [n_cold] where the row kernels leave its q: (cold row of its tile) * 64 + lane (CmView::src)
  int64_t cold_rows = 0;         // rows that may hold cold observations (rows >= tile.z of every tile), numbered tile after tile
  std::vector<int2> cold_range;  // [n_cams] run of each camera in the cold view
  int64_t rows = 0;
  int n_part_rec = 0, max_slots = 0, n_global = 0, n_tail = 0, grid_a = 1, grid_b = 1;
  int hubs = 0;      // leading slots of every workgroup with four accumulator replicas (V2::hubs)
  int strategy = 0;  // 0: rank-based camera grid, 1: contiguous landmark ranges with per-workgroup camera sets
};

// CPUs this process may actually use: the hardware threads, cut by the cgroup CPU quota when there is one (a container
// that sees 256 hardware threads under a 16-CPU quota gets slower, not faster, beyond 16 busy threads)
inline int lpl_effective_cpus() {
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    char q[32] = {0};
    long period = 0;
    if (std::fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0)
      n = std::min<long>(n, std::max<long>(1, (std::atol(q) + period - 1) / period));
    std::fclose(f);
  } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    long quota = -1, period = 0;
    if (std::fscanf(g, "%ld", &quota) != 1) quota = -1;
    std::fclose(g);
    if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (std::fscanf(h, "%ld", &period) != 1) period = 0;
      std::fclose(h);
    }
    if (quota > 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (quota + period - 1) / period));
  }
  return n;
}

// fn(item) for item in [0, n_items) on up to n_threads host threads (items taken on demand; the caller's thread works too)
// (cancel: when it turns true the remaining items are dropped -- a build whose result nobody waits for any more)
template <class F>
inline void lpl_parallel(int n_items, int n_threads, F&& fn, const std::atomic<bool>* cancel = nullptr) {
  std::atomic<int> next{0};
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n_items || (cancel && cancel->load(std::memory_order_relaxed))) break;
      fn(i);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < std::min(n_threads, n_items); ++t) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
}

// Minimum-cost assignment of h items to h positions (Hungarian algorithm, O(h^3); h is the number of resident
// observations of one landmark: 2..8 for almost all of them).  assign[item] = position.
inline void lpl_assign(int h, const std::vector<long>& cost, std::vector<int>& assign) {
  assign.resize(h);
  if (h == 1) { assign[0] = 0; return; }
  if (h == 2) {  // the two permutations (ties: identity, as the general algorithm below resolves them)
    const bool swap = cost[1] + cost[2] < cost[0] + cost[3];
    assign[0] = swap ? 1 : 0;
    assign[1] = swap ? 0 : 1;
    return;
  }
  const long INF = 1L << 60;
  // no heap traffic on the hot path (a million landmarks, several calls each): h <= 64 always (tile height <= 48,
  // a landmark dealt over lanes is only re-ordered up to 64 observations)
  constexpr int HMAX = 65;
  long u_s[HMAX], v_s[HMAX], minv_s[HMAX];
  int p_s[HMAX], way_s[HMAX];
  char used_s[HMAX];
  std::vector<long> heap_l;
  std::vector<int> heap_i;
  std::vector<char> heap_c;
  long *u = u_s, *v = v_s, *minv = minv_s;
  int *p = p_s, *way = way_s;
  char* used = used_s;
  if (h + 1 > HMAX) {
    heap_l.assign(3 * (size_t)(h + 1), 0);
    heap_i.assign(2 * (size_t)(h + 1), 0);
    heap_c.assign(h + 1, 0);
    u = heap_l.data(); v = u + h + 1; minv = v + h + 1;
    p = heap_i.data(); way = p + h + 1;
    used = heap_c.data();
  }
  for (int j = 0; j <= h; ++j) { u[j] = 0; v[j] = 0; p[j] = 0; way[j] = 0; }
  for (int i = 1; i <= h; ++i) {
    p[0] = i;
    int j0 = 0;
    for (int j = 0; j <= h; ++j) { minv[j] = INF; used[j] = 0; }
    do {
      used[j0] = 1;
      const int i0 = p[j0];
      long delta = INF;
      int j1 = 0;
      const long* crow = cost.data() + (size_t)(i0 - 1) * h - 1;
      for (int j = 1; j <= h; ++j)
        if (!used[j]) {
          const long cur = crow[j] - u[i0] - v[j];
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
  for (int j = 1; j <= h; ++j) assign[p[j] - 1] = j - 1;
}

// rank1[c]: 1-based popularity rank of camera c; cnt_sorted[r]: observation count of the camera with rank r (0-based)
inline void build_lpl(int n_cams, int n_lms, const int32_t* lm_off, const int32_t* cam_idx, const double* obs,
                      const std::vector<int>& rank1, const std::vector<int>& slot_of_obs, size_t n_slots, int grid,
                      int n_acc, LplLayout& L, bool place = true, const std::atomic<bool>* cancel = nullptr) {
  // cancel: povar_destroy does not wait for a placement it will not use (the layout is incomplete when it was set)
  auto cancelled = [&]() { return cancel && cancel->load(std::memory_order_relaxed); };
  // Rows per tile are capped by dealing longer landmarks over several lanes (K0 rows: at most 2 K0 row steps per
  // tile, the unit of load balance).  A wavefront walks its tile's rows one after the other, so on a problem too small
  // to give every wavefront a tile the cap sets the latency of the launch (ladybug-49: 2 rows 106 k, 8 rows 76 k
  // terms/s).
  grid = std::max(grid, 1);
  // every phase below is cut into independent pieces (landmark chunks, workgroups); the result does not depend on the
  // thread count
  int n_threads = std::min(lpl_effective_cpus(), 128);
  if (const char* e = std::getenv("POVAR_LAYOUT_THREADS")) n_threads = std::max(1, std::atoi(e));
  const bool timing = std::getenv("POVAR_LAYOUT_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[build_lpl] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // The cap follows the row steps a wavefront gets (observations / 64 lanes / 16 wavefronts per workgroup): a tile
  // longer than a wavefront's fair share is the tail of the launch, shorter ones cost lanes (landmark records,
  // segment sums) and leave the row placement fewer rows to dodge bank collisions with (venice-1778: 8 rows 13.8k,
  // 20-24 rows 14.2k, 32 rows 13.4k terms/s; its 8-GPU shards and the small shapes want 2)
  const long rows_per_wave = (long)(lm_off[n_lms] / WAVE) / ((long)grid * 16);
  int K0 = (int)std::min<long>(24, std::max<long>(2, rows_per_wave));
  if (const char* e = std::getenv("POVAR_LPL_K0")) K0 = std::max(2, std::atoi(e));
  // ---- grid factorisation and the (G, Tn) choice
  int B = 1;
  for (int b = 1; (int64_t)b * b <= grid; ++b)
    if (grid % b == 0) B = b;
  const int A = grid / B;
  std::vector<int64_t> S(n_cams + 1, 0);  // prefix sums of the observation counts in popularity order
  const int64_t n_obs_all = lm_off[n_lms];
  // landmark chunks of about equal observation counts: the unit of the parallel passes over the observations
  const int n_chunks = std::max(1, std::min(4 * n_threads, n_lms / 256 + 1));
  std::vector<int> chunk_lm(n_chunks + 1, n_lms);
  chunk_lm[0] = 0;
  for (int q = 1; q < n_chunks; ++q)
    chunk_lm[q] = (int)(std::upper_bound(lm_off, lm_off + n_lms + 1, (int32_t)(n_obs_all * q / n_chunks)) - lm_off) - 1;
  for (int q = 1; q <= n_chunks; ++q) chunk_lm[q] = std::max(chunk_lm[q], chunk_lm[q - 1]);
  {
    std::vector<std::vector<int64_t>> cnt_q(n_chunks);
    lpl_parallel(n_chunks, n_threads, [&](int q) {
      cnt_q[q].assign(n_cams, 0);
      for (int64_t i = lm_off[chunk_lm[q]]; i < lm_off[chunk_lm[q + 1]]; ++i) cnt_q[q][rank1[cam_idx[i]] - 1]++;
    });
    for (int r = 0; r < n_cams; ++r) {
      int64_t c = 0;
      for (int q = 0; q < n_chunks; ++q) c += cnt_q[q][r];
      S[r + 1] = S[r] + c;
    }
  }
  int G = std::min(n_cams, n_acc), Tn = 0;
  double cov_grid = (double)n_obs_all;  // observations the chosen (G, Tn) keeps LDS-resident (counted estimate)
  if (n_cams > n_acc && std::getenv("POVAR_LPL_NOGRID") == nullptr) {
    // Candidates g (global cameras) with the grid cameras the remaining slots allow; the cover of each is COUNTED per
    // landmark: its observations of global cameras, its first two grid observations (always resident, whichever the
    // cameras), and every further grid observation with the probability that its camera happens to sit in the chosen
    // workgroup's column or row.  (Round 2 scored the grid observations with a flat 0.85: right for venice-1778,
    // 5 observations per landmark, too optimistic for final-13682 with 6.5 -- it took 84 global cameras where 240 leave
    // 23.5 % instead of 26.2 % of the observations cold.)
    std::vector<int> cand_g, cand_tn;
    for (int g = 0; g <= n_acc; g += 20) {
      const int cap = n_acc - g;
      // slots a workgroup needs: its column (<= Tn/A + 1 cameras) and its row (<= Tn/B + A: blocks of A ranks)
      int tn = (int)std::max(0.0, (cap - A - 2) / (1.0 / A + 1.0 / B));
      cand_g.push_back(g);
      cand_tn.push_back(std::min(tn, n_cams - g));
    }
    const int n_cand = (int)cand_g.size();
    const double p_extra = 1.0 / A + 1.0 / B - 1.0 / ((double)A * B);
    // g_k ascends and g_k + tn_k descends with k, so a camera of rank r is global for the candidates k >= kg[r] and a
    // grid camera for k < min(kg[r], kt[r]): two table lookups per observation, the counts per candidate by prefix sums
    std::vector<int> kg(n_cams), kt(n_cams);
    for (int r = 0; r < n_cams; ++r) {
      int a = 0;
      while (a < n_cand && cand_g[a] <= r) ++a;
      kg[r] = a;
      int b = 0;
      while (b < n_cand && r < cand_g[b] + cand_tn[b]) ++b;
      kt[r] = b;
    }
    std::vector<std::vector<double>> cov_q(n_chunks, std::vector<double>(n_cand, 0.0));
    lpl_parallel(n_chunks, n_threads, [&](int q) {
      std::vector<double>& cov = cov_q[q];
      std::vector<int> dg(n_cand + 1), dt(n_cand + 1);
      for (int l = chunk_lm[q]; l < chunk_lm[q + 1]; ++l) {
        std::fill(dg.begin(), dg.end(), 0);
        std::fill(dt.begin(), dt.end(), 0);
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
          const int r0 = rank1[cam_idx[i]] - 1;
          dg[kg[r0]]++;                       // global for k >= kg
          dt[std::min(kg[r0], kt[r0])]++;     // grid for k < min(kg, kt)
        }
        int ng = 0, nt = lm_off[l + 1] - lm_off[l];  // nt: observations that are grid cameras for candidate k
        int below = 0;
        for (int k = 0; k < n_cand; ++k) {
          ng += dg[k];
          below += dt[k];                     // observations whose grid range ended before k
          const int ntk = nt - below;
          cov[k] += ng + std::min(ntk, 2) + p_extra * std::max(ntk - 2, 0);
        }
      }
    });
    double best = -1;
    for (int k = 0; k < n_cand; ++k) {
      double c = 0;
      for (int q = 0; q < n_chunks; ++q) c += cov_q[q][k];
      if (c > best) { best = c; G = cand_g[k]; Tn = cand_tn[k]; }
    }
    cov_grid = best;
    if (const char* e = std::getenv("POVAR_LPL_G")) {  // measurement knob: force the number of global cameras
      G = std::min(std::max(0, std::atoi(e)), n_acc);
      Tn = std::min((int)std::max(0.0, (n_acc - G - A - 2) / (1.0 / A + 1.0 / B)), n_cams - G);
    }
  }
  // ---- the alternative for graphs with locality: contiguous landmark ranges, each workgroup keeping the cameras ITS
  // landmarks observe most.  The reference gets its locality from the landmark order of the file
  // (bal/bal_problem.cpp:183-303); in a real reconstruction neighbouring landmarks share cameras, so a range of
  // 1 / grid of the landmarks touches a few hundred cameras and nearly all of its observations become LDS-resident,
  // whatever the global popularity law.  On a graph WITHOUT locality (the SURVEY 8(d) Zipf workload: every landmark
  // samples the whole camera set) a range sees every camera and the rank-based grid above covers more.  Both covers
  // are counted and the better one is taken (POVAR_LPL_STRATEGY=grid|range forces one).
  std::vector<int> range_lm(grid + 1, n_lms);
  range_lm[0] = 0;
  for (int w = 1; w < grid; ++w)
    range_lm[w] = (int)(std::upper_bound(lm_off, lm_off + n_lms + 1, (int32_t)(n_obs_all * w / grid)) - lm_off) - 1;
  for (int w = 1; w <= grid; ++w) range_lm[w] = std::max(range_lm[w], range_lm[w - 1]);
  const int hubs_r = lpl_hubs(std::min(n_cams, n_acc));
  std::vector<std::vector<int>> range_set(grid);   // resident ranks of workgroup w, ascending (range strategy)
  std::vector<std::vector<int>> range_hubs(grid);  // its hubs_r most observed ones: the slots with accumulator replicas
  bool use_range = false;
  if (n_cams > n_acc) {
    std::vector<int64_t> cov(grid, 0);
    lpl_parallel(grid, n_threads, [&](int w) {
      std::vector<int> cnt(n_cams, 0);
      for (int64_t i = lm_off[range_lm[w]]; i < lm_off[range_lm[w + 1]]; ++i) cnt[rank1[cam_idx[i]] - 1]++;
      std::vector<int> cand;
      for (int r0 = 0; r0 < n_cams; ++r0)
        if (cnt[r0] > 0) cand.push_back(r0);
      const size_t keep = std::min(cand.size(), (size_t)n_acc);
      std::partial_sort(cand.begin(), cand.begin() + keep, cand.end(),
                        [&](int a, int b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
      cand.resize(keep);
      int64_t c = 0;
      for (int r0 : cand) c += cnt[r0];
      range_hubs[w].assign(cand.begin(), cand.begin() + std::min(cand.size(), (size_t)hubs_r));
      // a workgroup with fewer cameras than hub slots fills them with unused ranks (the slots must exist)
      for (int r0 = 0; (int)range_hubs[w].size() < hubs_r && r0 < n_cams; ++r0)
        if (std::find(cand.begin(), cand.end(), r0) == cand.end() && (int)cand.size() < n_acc) {
          range_hubs[w].push_back(r0);
          cand.push_back(r0);
        }
      std::sort(cand.begin(), cand.end());
      range_set[w] = cand;
      cov[w] = c;
    });
    int64_t cov_range = 0;
    for (int w = 0; w < grid; ++w) cov_range += cov[w];
    use_range = (double)cov_range > cov_grid;
    if (const char* e = std::getenv("POVAR_LPL_STRATEGY")) use_range = e[0] == 'r';
    if (timing) std::fprintf(stderr, "[build_lpl] cover: ranges %.4f, grid (estimate) %.4f -> %s\n", (double)cov_range / n_obs_all,
                             cov_grid / n_obs_all, use_range ? "ranges" : "grid");
  }
  if (use_range) {
    G = 0;  // nothing is resident everywhere: every workgroup has its own camera set, hubs included
    Tn = 0;
  }
  L.strategy = use_range ? 1 : 0;
  L.hubs = use_range ? hubs_r : lpl_hubs(G);
  L.n_global = G;
  L.n_tail = Tn;
  L.grid_a = A;
  L.grid_b = B;
  // grid coordinates of every camera by rank (-1: not a grid camera)
  std::vector<int16_t> tx(n_cams, -1), ty(n_cams, -1);
  for (int t = 0; t < Tn; ++t) {
    tx[G + t] = (int16_t)(t % A);
    ty[G + t] = (int16_t)((t / A) % B);
  }
  auto tail_xy = [&](int r0, int& x, int& y) {  // r0: 0-based rank; false if not a grid camera
    x = tx[r0];
    y = ty[r0];
    return x >= 0;
  };
  auto resident = [&](int w, int r0) {
    if (r0 < G) return true;
    if (use_range) return std::binary_search(range_set[w].begin(), range_set[w].end(), r0);
    return tx[r0] >= 0 && (tx[r0] == w % A || ty[r0] == w / A);
  };
  lap("grid choice");
  if (cancelled()) return;
  // ---- landmark -> workgroup
  std::vector<int> wg_of(n_lms, -1);
  std::vector<int64_t> load(grid, 0);
  if (use_range) {
    lpl_parallel(grid, n_threads, [&](int w) {
      for (int l = range_lm[w]; l < range_lm[w + 1]; ++l)
        if (lm_off[l + 1] > lm_off[l]) wg_of[l] = w;
    });
  } else {
    std::vector<int> n_tail_of(n_lms, 0), order;
    lpl_parallel(n_chunks, n_threads, [&](int q) {
      for (int l = chunk_lm[q]; l < chunk_lm[q + 1]; ++l) {
        int n = 0;
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) n += tx[rank1[cam_idx[i]] - 1] >= 0;
        n_tail_of[l] = n;
      }
    });
    {
      // most grid observations first, landmark order inside a count (a stable counting sort)
      int mx = 0;
      for (int l = 0; l < n_lms; ++l) mx = std::max(mx, n_tail_of[l]);
      std::vector<int> first(mx + 2, 0);
      for (int l = 0; l < n_lms; ++l)
        if (lm_off[l + 1] > lm_off[l]) first[mx - n_tail_of[l] + 1]++;
      for (int v = 0; v <= mx; ++v) first[v + 1] += first[v];
      order.resize(first[mx + 1]);
      for (int l = 0; l < n_lms; ++l)
        if (lm_off[l + 1] > lm_off[l]) order[first[mx - n_tail_of[l]]++] = l;
    }
    // Landmarks with grid observations: most constrained first, each to the eligible workgroup that covers most of its
    // grid cameras (ties: the least loaded).  The only coupling between landmarks is the load balance, so the list is
    // dealt round-robin over a FIXED number of independent lanes (64, whatever the thread count: the layout must not
    // depend on it), each balancing its own share; the shares are equal mixes of all constraint levels.
    size_t pos = 0;
    while (pos < order.size() && n_tail_of[order[pos]] > 0) ++pos;
    {
      constexpr int LANES = 64;
      std::vector<std::vector<int64_t>> load_q(LANES, std::vector<int64_t>(grid, 0));
      lpl_parallel(LANES, n_threads, [&](int q) {
        std::vector<int64_t>& ld = load_q[q];
        std::vector<int> xs, ys;
        for (size_t n = q; n < pos; n += LANES) {
          const int l = order[n], k = lm_off[l + 1] - lm_off[l];
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
            if (cov > best_cov || (cov == best_cov && ld[w] < ld[best_w])) { best_cov = cov; best_w = w; }
          };
          if (nt == 1) {  // any workgroup of the column or the row
            for (int b = 0; b < B; ++b) consider(xs[0], b);
            for (int a = 0; a < A; ++a) consider(a, ys[0]);
          } else {
            for (int i = 0; i < nc; ++i)
              for (int j = 0; j < nc; ++j) consider(xs[i], ys[j]);
          }
          wg_of[l] = best_w;
          ld[best_w] += k;
        }
      });
      for (int q = 0; q < LANES; ++q)
        for (int w = 0; w < grid; ++w) load[w] += load_q[q][w];
    }
    // landmarks without tail observations fill the workgroups up, longest first into the least loaded
    std::priority_queue<std::pair<int64_t, int>, std::vector<std::pair<int64_t, int>>, std::greater<>> heap;
    for (int w = 0; w < grid; ++w) heap.push({load[w], w});
    std::vector<int> rest(order.size() - pos);
    {
      // longest first, landmark order inside a length (stable counting sort)
      int mx = 0;
      for (size_t n = pos; n < order.size(); ++n) mx = std::max(mx, (int)(lm_off[order[n] + 1] - lm_off[order[n]]));
      std::vector<int> first(mx + 2, 0);
      for (size_t n = pos; n < order.size(); ++n) first[mx - (lm_off[order[n] + 1] - lm_off[order[n]]) + 1]++;
      for (int v = 0; v <= mx; ++v) first[v + 1] += first[v];
      for (size_t n = pos; n < order.size(); ++n) rest[first[mx - (lm_off[order[n] + 1] - lm_off[order[n]])]++] = order[n];
    }
    for (int l : rest) {
      auto [ld, w] = heap.top();
      heap.pop();
      wg_of[l] = w;
      load[w] = ld + (lm_off[l + 1] - lm_off[l]);
      heap.push({load[w], w});
    }
  }
  lap("landmark -> workgroup");
  if (cancelled()) return;
  // ---- cold view: observations whose camera is not resident in their landmark's workgroup, camera-major
  const int64_t n_obs = lm_off[n_lms];
  std::vector<int> cold_pos_of_obs(n_obs, -1);
  {
    // per (chunk, camera) counts -> every chunk fills its own sub-run of a camera's run: the order inside a camera is
    // the landmark order, whatever the thread count
    std::vector<std::vector<int>> cq(n_chunks);
    lpl_parallel(n_chunks, n_threads, [&](int q) {
      std::vector<int>& c = cq[q];
      c.assign(n_cams, 0);
      for (int l = chunk_lm[q]; l < chunk_lm[q + 1]; ++l)
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i)
          if (!resident(wg_of[l], rank1[cam_idx[i]] - 1)) c[cam_idx[i]]++;
    });
    L.cold_range.resize(n_cams);
    int run = 0;
    for (int c = 0; c < n_cams; ++c) {
      const int first = run;
      for (int q = 0; q < n_chunks; ++q) {
        const int n = cq[q][c];
        cq[q][c] = run;  // where chunk q starts inside camera c's run
        run += n;
      }
      L.cold_range[c] = make_int2(first, run);
    }
    L.cold_lm.resize(run);
    lpl_parallel(n_chunks, n_threads, [&](int q) {
      std::vector<int>& fill = cq[q];
      for (int l = chunk_lm[q]; l < chunk_lm[q + 1]; ++l)
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i)
          if (!resident(wg_of[l], rank1[cam_idx[i]] - 1)) {
            const int p = fill[cam_idx[i]]++;
            cold_pos_of_obs[i] = p;
            L.cold_lm[p] = l;
          }
    });
  }
  lap("cold view");
  if (cancelled()) return;
  // ---- per workgroup: camera slots (global cameras in rank order, then the tail cameras its landmarks use), tiles
  std::vector<std::vector<int>> lms_of(grid);
  for (int l = 0; l < n_lms; ++l)
    if (wg_of[l] >= 0) lms_of[wg_of[l]].push_back(l);
  lap("  lms_of lists");
  L.wg_cam_off.assign(grid + 1, 0);
  L.wg_tile_off.assign(grid + 1, 0);
  L.lm_pos.assign(n_lms, -1);
  L.of_slot.assign(n_slots, -1);
  std::vector<std::vector<int>> holders(n_cams);  // rank -> workgroups with a slot for it (for the partial records)
  // per workgroup, aligned with its lane order order_of[w] (workgroup-local arrays: the landmarks of a workgroup are
  // spread over the whole index range, per-landmark arrays shared by the threads would bounce between their caches)
  struct LmInfo { int l, parts, psize, cold, pos; };
  std::vector<std::vector<LmInfo>> info_of(grid);
  std::vector<std::vector<int>> order_of(grid);
  {
    // every workgroup on its own (parallel): slots, lane order, tiles with workgroup-local tile numbers ...
    struct WgOut {
      std::vector<int> cams, seg, lm_of;
      std::vector<int4> tile;
    };
    std::vector<WgOut> out(grid);
    lpl_parallel(grid, n_threads, [&](int w) {
      WgOut& o = out[w];
      // slots: the replicated hub cameras always (fixed slots 0..hubs-1), then only the resident cameras this
      // workgroup's landmarks actually observe, in rank order (a small shard touches far fewer than G + its grid
      // cameras: fewer records to stage, fewer partial records to flush and to sum)
      const int hubs_w = L.hubs;
      std::vector<long> wcnt(n_cams, 0);  // this workgroup's observations per camera (by rank)
      for (int l : lms_of[w])
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) wcnt[rank1[cam_idx[i]] - 1]++;
      // hub slots: the cameras with four accumulator replicas -- the most observed ones overall (grid strategy: the
      // same in every workgroup) or of this workgroup (range strategy)
      std::vector<int> hub_cams;
      if (use_range) hub_cams = range_hubs[w];
      else for (int r0 = 0; r0 < hubs_w; ++r0) hub_cams.push_back(r0);
      std::vector<char> is_hub(n_cams, 0);
      for (int r0 : hub_cams) is_hub[r0] = 1;
      std::vector<int> used;
      for (int r0 = 0; r0 < n_cams; ++r0)
        if (!is_hub[r0] && wcnt[r0] > 0 && resident(w, r0)) used.push_back(r0);
      for (int r0 : hub_cams) o.cams.push_back(r0);
      // Which slot a camera gets decides its LDS bank (accumulators: (slot + 3 hubs) mod 32, records: slot mod 16).
      // In rank order the popular cameras pile up on the low banks and force collisions no row placement can avoid
      // (a bank hit by more observations than the tile has rows must repeat inside a row).  So the cameras are dealt
      // to the banks heaviest first, each to the least loaded bank that still has a free slot (weights = this
      // workgroup's observations per camera; the hubs' four replicas are pre-loaded with a quarter each).
      {
        const int n_rest = (int)used.size(), n_w_ = hubs_w + n_rest;
        std::vector<double> load(32, 0.0);
        for (int hs = 0; hs < hubs_w; ++hs)
          for (int q = 0; q < 4; ++q) load[(4 * hs + q) & 31] += 0.25 * (double)wcnt[hub_cams[hs]];
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
        for (int sl = hubs_w; sl < n_w_; ++sl) o.cams.push_back(cam_of_slot[sl]);
      }
      // tiles: lanes sorted by (rows per lane, cold rows per lane), longest first
      std::vector<int>& order = order_of[w];
      std::vector<LmInfo>& info = info_of[w];
      for (int l : lms_of[w]) {
        const int k = lm_off[l + 1] - lm_off[l];
        int cold = 0;
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) cold += cold_pos_of_obs[i] >= 0;
        LmInfo f;
        f.l = l;
        f.parts = k <= K0 ? 1 : std::min(WAVE, (k + K0 - 1) / K0);
        f.psize = (k + f.parts - 1) / f.parts;
        f.cold = (cold + f.parts - 1) / f.parts;
        f.pos = 0;
        info.push_back(f);
      }
      std::stable_sort(info.begin(), info.end(), [&](const LmInfo& a, const LmInfo& b) {
        return a.psize != b.psize ? a.psize > b.psize : a.cold > b.cold;
      });
      int tile = 0, fill = 0;
      for (LmInfo& f : info) {
        order.push_back(f.l);
        if (fill + f.parts > WAVE) { ++tile; fill = 0; }
        f.pos = tile * WAVE + fill;  // local tile number: rebased below
        fill += f.parts;
      }
      const int n_tiles_w = order.empty() ? 0 : tile + 1;
      o.tile.assign(n_tiles_w, make_int4(0, 0, 1 << 30, 0));
      o.seg.resize((size_t)n_tiles_w * WAVE);
      o.lm_of.assign((size_t)n_tiles_w * WAVE, -1);
      for (size_t i = 0; i < o.seg.size(); ++i) o.seg[i] = (int)(i & 63) | ((int)(i & 63) << 8);
      std::vector<int> lanes_used(n_tiles_w, 0);
      for (const LmInfo& f : info) {
        const int l = f.l, pos = f.pos, t = pos >> 6, lane0 = pos & 63, P = f.parts;
        int hot = 0;
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) hot += cold_pos_of_obs[i] < 0;
        int4& ti = o.tile[t];
        ti.y = std::max(ti.y, f.psize);
        ti.z = std::min(ti.z, hot / P);  // leading rows in which every lane of the landmark has a resident camera
        if (P > 1) ti.w |= 1;
        for (int q = 0; q < P; ++q) {
          o.seg[(size_t)t * WAVE + lane0 + q] = lane0 | ((lane0 + P - 1) << 8);
          o.lm_of[(size_t)t * WAVE + lane0 + q] = l;
        }
        lanes_used[t] += P;
      }
      for (int t = 0; t < n_tiles_w; ++t) {
        if (lanes_used[t] < WAVE) o.tile[t].z = 0;  // unused lanes: no branch-free rows
        // at least two rows = four row steps per tile: the prefetch cursor (three rows ahead) then never needs a
        // tile beyond the one the consumer has already taken
        o.tile[t].y = std::max(o.tile[t].y, 2);
      }
    });
    lap("  slots+tiles parallel part");
    // ... then the workgroups one after the other: global slot / tile / row numbers
    std::vector<int> tile0_of(grid, 0);
    for (int w = 0; w < grid; ++w) {
      const WgOut& o = out[w];
      const int slot0 = (int)L.wg_cams.size(), n_w = (int)o.cams.size();
      L.wg_cams.insert(L.wg_cams.end(), o.cams.begin(), o.cams.end());
      L.max_slots = std::max(L.max_slots, n_w);
      for (int s = 0; s < n_w; ++s) holders[L.wg_cams[slot0 + s]].push_back(w);
      L.wg_cam_off[w + 1] = (int)L.wg_cams.size();
      tile0_of[w] = (int)L.tile.size();
      for (int4 ti : o.tile) {
        ti.x = (int)L.rows;
        L.rows += ti.y;
        // bits 4.. of the flag word: number of the tile's first cold row (a cold observation in row j >= tile.z leaves
        // its q at ((w >> 4) + j - z) * 64 + lane: the lanes of a wavefront store side by side, the per-camera kernel
        // gathers -- a scatter of 32-byte stores from the row kernel cost final-13682 a third of its term)
        ti.z = std::min(ti.z, ti.y);
        ti.w |= (int)(L.cold_rows << 4);
        L.cold_rows += ti.y - ti.z;
        L.tile.push_back(ti);
      }
      L.seg.insert(L.seg.end(), o.seg.begin(), o.seg.end());
      L.lm_of.insert(L.lm_of.end(), o.lm_of.begin(), o.lm_of.end());
      L.wg_tile_off[w + 1] = (int)L.tile.size();
    }
    lpl_parallel(grid, n_threads, [&](int w) {
      for (LmInfo& f : info_of[w]) f.pos += tile0_of[w] * WAVE;
    });
  }
  lap("slots and tiles");
  if (cancelled()) return;
  L.uv.assign((size_t)L.rows * WAVE, make_double2(0, 0));
  L.cw.assign((size_t)L.rows * WAVE, -1);
  L.cpos.assign((size_t)L.rows * WAVE, -1);
  L.cold_src.assign(L.cold_lm.size(), -1);
  // ---- row stream with LDS bank placement, workgroups in parallel (disjoint row ranges).
  // ds_add_f64 takes 8 LDS cycles per wavefront when the 32 lanes of each half hit 32 different bank pairs and 8
  // more for every additional lane on a bank (tools/micro/lds_atomic_rates.hip); a ds_read_b128 takes one more
  // cycle per lane group for every additional record on a bank quad.  With the observations in their natural order
  // row 0 would hold every landmark's lowest-index camera.  So the assignment of a landmark's resident
  // observations to the rows of its lanes is an assignment problem against the banks already taken in its tile
  // (accumulator bank = lpl_acc_slot(slot, lane) mod 32 per row half, record quad = slot mod 16 per read group):
  // solved exactly per landmark (Hungarian), landmark after landmark, then once more with everything placed.  The
  // single-lane landmarks of one class are interchangeable, so the greedy also picks the tile and the lane.
  auto read_group = [](int lane) {
    const int l = lane & 31;
    const int g = (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1;
    return g + 2 * (lane >> 5);
  };
  const int hubs = L.hubs;
  // place = false: rows in their natural order.  Everything but the six row-order arrays (uv, cw, cpos, lm_pos, lm_of,
  // of_slot) is the same as with placement: povar_create starts on the natural order and swaps the placed rows in when
  // a host thread has finished them (lpl_row_arrays_only_differ in the layout checker holds the two against each other)
  const bool no_place = !place;
  int max_tiles_tried = 1 << 30;
  if (const char* e = std::getenv("POVAR_LPL_TILES_TRIED")) max_tiles_tried = std::max(1, std::atoi(e));
  auto place_wg = [&](int w) {
    std::vector<int> slot_of_rank(n_cams, -1);
    std::vector<int> hot_idx, cold_idx, assign, assign2;
    std::vector<char> rep_tmp;
    std::vector<long> cost;
    {
      const int slot0 = L.wg_cam_off[w], n_w = L.wg_cam_off[w + 1] - slot0;
      for (int s = 0; s < n_w; ++s) slot_of_rank[L.wg_cams[slot0 + s]] = s;
      const std::vector<int>& order = order_of[w];
      const std::vector<LmInfo>& info = info_of[w];
      const int t0w = L.wg_tile_off[w], ntw = L.wg_tile_off[w + 1] - t0w;
      // bank occupancy of every tile of the workgroup: [tile][row][2 halves x 32 | 4 groups x 16]
      std::vector<size_t> occ_off(ntw + 1, 0);
      for (int t = 0; t < ntw; ++t) occ_off[t + 1] = occ_off[t] + (size_t)L.tile[t0w + t].y * 64;
      std::vector<uint16_t> occA(occ_off[ntw], 0), occR(occ_off[ntw], 0);
      // hub records are read too: all lanes of a read group that want the same hub are one broadcast access, the
      // first of them puts the hub's record on its read class (hub mod 16); cntH counts them per (tile, row, group, hub)
      std::vector<uint16_t> cntH(occ_off[ntw], 0);
      auto idxH = [&](int t, int hub, int lane, int j) { return occ_off[t - t0w] + (size_t)j * 64 + read_group(lane) * 16 + (hub & 15); };
      auto idxA = [&](int t, int lane, int j, int acc_slot) { return occ_off[t - t0w] + (size_t)j * 64 + (lane >> 5) * 32 + (acc_slot & 31); };
      auto idxR = [&](int t, int sl, int lane, int j) { return occ_off[t - t0w] + (size_t)j * 64 + read_group(lane) * 16 + (sl & 15); };
      // What a collision costs is the WORST multiplicity of its row (per half for ds_add_f64, per lane group for
      // ds_read_b128): a second collision in a row that already has one is free.  So the cost of putting an
      // observation on a bank is the increase of that maximum (ties: the bank's occupancy), which makes the greedy
      // gather the unavoidable collisions in few rows instead of spreading one into every row.
      std::vector<uint16_t> mxA(occ_off[ntw] / 32, 1), mxR(occ_off[ntw] / 16, 1);  // per (tile, row, half) / (tile, row, group)
      auto mxA_at = [&](int t, int lane, int j) -> uint16_t& { return mxA[(occ_off[t - t0w] + (size_t)j * 64) / 32 + (lane >> 5)]; };
      auto mxR_at = [&](int t, int lane, int j) -> uint16_t& { return mxR[(occ_off[t - t0w] + (size_t)j * 64) / 16 + read_group(lane)]; };
      // cost of an observation of slot sl at (t, lane, row j); hubs pick their best replica
      auto occ_of = [&](int t, int sl, int lane, int j, int& rep) -> long {
        const int m = mxA_at(t, lane, j);
        if (sl >= hubs) {
          rep = 0;
          const int v = occA[idxA(t, lane, j, sl + 3 * hubs)];
          return 1000L * std::max(0, v + 1 - m) + v;
        }
        long best = 1L << 40;
        for (int q = 0; q < 4; ++q) {
          const int v = occA[idxA(t, lane, j, 4 * sl + q)];
          const long c = 1000L * std::max(0, v + 1 - m) + v;
          if (c < best) { best = c; rep = q; }
        }
        return best;
      };
      auto read_cost = [&](int t, int sl, int lane, int j) -> long {
        if (sl < hubs && cntH[idxH(t, sl, lane, j)] > 0) return 0;  // joins a broadcast
        const int v = occR[idxR(t, sl, lane, j)];
        return 1000L * std::max(0, v + 1 - (int)mxR_at(t, lane, j)) + v;
      };
      const size_t n_o = order.size();
      // per landmark: its resident observations in position order (one flat array per workgroup, no per-landmark
      // allocations), the LDS slot of each (looked up once) and the accumulator replica of each (hubs)
      std::vector<int> placed_off(n_o + 1, 0), tile_of(n_o), lane_of(n_o);
      for (size_t o = 0; o < n_o; ++o) {
        const int pos = info[o].pos;
        tile_of[o] = pos >> 6;
        lane_of[o] = pos & 63;
        int hot = 0;
        for (int i = lm_off[order[o]]; i < lm_off[order[o] + 1]; ++i) hot += cold_pos_of_obs[i] < 0;
        placed_off[o + 1] = placed_off[o] + hot;
      }
      std::vector<int> placed_obs(placed_off[n_o]), placed_slot(placed_off[n_o]);
      std::vector<char> placed_rep(placed_off[n_o], 0);
      for (size_t o = 0; o < n_o; ++o) {
        int n = placed_off[o];
        for (int i = lm_off[order[o]]; i < lm_off[order[o] + 1]; ++i)
          if (cold_pos_of_obs[i] < 0) {
            placed_obs[n] = i;
            placed_slot[n] = slot_of_rank[rank1[cam_idx[i]] - 1];
            ++n;
          }
      }
      struct Placed {  // view of one landmark's run
        int *obs, *slot;
        char* rep;
        int n;
        int size() const { return n; }
      };
      auto placed = [&](size_t o) { return Placed{placed_obs.data() + placed_off[o], placed_slot.data() + placed_off[o],
                                                   placed_rep.data() + placed_off[o], placed_off[o + 1] - placed_off[o]}; };
      // cheapest assignment of the landmark's resident observations to the rows of lanes [lane0, lane0 + P) of tile t
      auto place_cost = [&](const Placed& cur, int t, int lane0, int P, std::vector<int>& out_assign) -> long {
        const int h = cur.size();
        if (h == 0) return 0;
        cost.assign((size_t)h * h, 0);
        for (int a = 0; a < h; ++a) {
          const int sl = cur.slot[a];
          for (int pos = 0; pos < h; ++pos) {
            const int lane = lane0 + pos % P, j = pos / P;
            int rp;
            long c = 6L * occ_of(t, sl, lane, j, rp);  // 12 atomics x 8 cycles against 16 reads x 1 cycle per extra lane
            c += read_cost(t, sl, lane, j);
            cost[(size_t)a * h + pos] = c;
          }
        }
        lpl_assign(h, cost, out_assign);
        long tot = 0;
        for (int a = 0; a < h; ++a) tot += cost[(size_t)a * h + out_assign[a]];
        return tot;
      };
      auto commit = [&](size_t o, int sign, bool choose_rep) {
        const Placed cur = placed(o);
        const int P = info[o].parts;
        for (int n = 0; n < cur.size(); ++n) {
          const int sl = cur.slot[n], lane = lane_of[o] + n % P, j = n / P, t = tile_of[o];
          int acc_slot = sl + 3 * hubs;
          if (sl < hubs) {
            int rp = cur.rep[n];
            if (choose_rep) { occ_of(t, sl, lane, j, rp); cur.rep[n] = (char)rp; }
            acc_slot = 4 * sl + rp;
          }
          uint16_t& ca = occA[idxA(t, lane, j, acc_slot)];
          ca += sign;
          if (sign > 0) mxA_at(t, lane, j) = std::max(mxA_at(t, lane, j), ca);
          else {  // removal: the row's maximum may drop
            uint16_t m = 1;
            const size_t base = occ_off[t - t0w] + (size_t)j * 64 + (lane >> 5) * 32;
            for (int q = 0; q < 32; ++q) m = std::max(m, occA[base + q]);
            mxA_at(t, lane, j) = m;
          }
          bool touches_class = sl >= hubs;
          if (sl < hubs) {  // the class sees a hub once per read group, however many lanes broadcast it
            uint16_t& ch = cntH[idxH(t, sl, lane, j)];
            ch += sign;
            touches_class = sign > 0 ? ch == 1 : ch == 0;
          }
          if (touches_class) {
            uint16_t& cr = occR[idxR(t, sl, lane, j)];
            cr += sign;
            if (sign > 0) mxR_at(t, lane, j) = std::max(mxR_at(t, lane, j), cr);
            else {
              uint16_t m = 1;
              const size_t base = occ_off[t - t0w] + (size_t)j * 64 + read_group(lane) * 16;
              for (int q = 0; q < 16; ++q) m = std::max(m, occR[base + q]);
              mxR_at(t, lane, j) = m;
            }
          }
        }
      };
      auto reorder = [&](size_t o) {
        const Placed cur = placed(o);
        const int h = cur.size();
        if (h <= 1) return;
        hot_idx.assign(2 * (size_t)h, 0);
        for (int a = 0; a < h; ++a) {
          hot_idx[assign[a]] = cur.obs[a];
          hot_idx[h + assign[a]] = cur.slot[a];
        }
        for (int a = 0; a < h; ++a) {
          cur.obs[a] = hot_idx[a];
          cur.slot[a] = hot_idx[h + a];
        }
      };
      if (!no_place) {
        // (1) landmarks dealt over several lanes keep their lanes: rows only
        for (size_t o = 0; o < n_o; ++o)
          if (info[o].parts > 1) {
            if (placed(o).size() <= 64) { place_cost(placed(o), tile_of[o], lane_of[o], info[o].parts, assign); reorder(o); }
            commit(o, +1, true);
          }
        // (2) single-lane landmarks of one class (same rows per lane, same cold rows) are interchangeable.
        size_t a = 0;
        while (a < n_o) {
          if (info[a].parts > 1) { ++a; continue; }
          size_t b = a;
          while (b < n_o && info[b].parts == 1 && info[b].psize == info[a].psize && info[b].cold == info[a].cold) ++b;
          const int ta = tile_of[a], tb = tile_of[b - 1];
          std::vector<unsigned long long> free_mask(tb - ta + 1, 0);
          for (size_t o = a; o < b; ++o) free_mask[tile_of[o] - ta] |= 1ull << lane_of[o];
          // each takes the cheapest free position of the class, one landmark after the other: any tile of the class,
          // any class of lanes (half, ds_read_b128 lane group), rows by the assignment problem
          for (size_t o = a; o < b; ++o) {
            long best = -1;
            int best_t = -1, best_lane = -1;
            int tiles_tried = 0;
            for (int t = ta; t <= tb && best != 0 && tiles_tried < max_tiles_tried; ++t) {
              if (!free_mask[t - ta]) continue;
              ++tiles_tried;
              unsigned seen = 0;
              for (unsigned long long m = free_mask[t - ta]; m && best != 0; m &= m - 1) {
                const int lane = __builtin_ctzll(m), cls = (lane >> 5) * 2 + (read_group(lane) & 1);
                if (seen & (1u << cls)) continue;
                seen |= 1u << cls;
                const long c = place_cost(placed(o), t, lane, 1, assign2);
                if (best < 0 || c < best) { best = c; best_t = t; best_lane = lane; assign = assign2; }
              }
            }
            tile_of[o] = best_t;
            lane_of[o] = best_lane;
            free_mask[best_t - ta] &= ~(1ull << best_lane);
            reorder(o);
            commit(o, +1, true);
          }
          a = b;
        }
        // (3) once more with everything placed: rows only (not the exactly placed ones)
        for (size_t o = 0; o < n_o; ++o) {
          if (placed(o).size() < 2 || placed(o).size() > 64) continue;
          commit(o, -1, false);
          place_cost(placed(o), tile_of[o], lane_of[o], info[o].parts, assign);
          reorder(o);
          commit(o, +1, true);
        }
      } else {
        for (size_t o = 0; o < n_o; ++o) commit(o, +1, true);
      }
      // write the rows
      for (size_t o = 0; o < n_o; ++o) {
        const int l = order[o], t = tile_of[o], lane0 = lane_of[o], P = info[o].parts;
        L.lm_pos[l] = (t * WAVE + lane0) | ((P - 1) << 26);
        if (P == 1) L.lm_of[(size_t)t * WAVE + lane0] = l;
        const Placed cur = placed(o);
        const int h = cur.size();
        cold_idx.clear();
        for (int i = lm_off[l]; i < lm_off[l + 1]; ++i)
          if (cold_pos_of_obs[i] >= 0) cold_idx.push_back(i);
        for (int n = 0; n < h + (int)cold_idx.size(); ++n) {
          const int i = n < h ? cur.obs[n] : cold_idx[n - h];
          const int q = n % P, j = n / P, r0 = rank1[cam_idx[i]] - 1, lane = lane0 + q;
          const size_t idx = ((size_t)L.tile[t].x + j) * WAVE + lane;
          L.uv[idx] = make_double2(obs[2 * (size_t)i], obs[2 * (size_t)i + 1]);
          L.of_slot[slot_of_obs[i]] = (int)idx;
          if (n < h) {
            L.cw[idx] = cur.slot[n] | ((int)cur.rep[n] << 16);
          } else {
            L.cw[idx] = -2 - r0;  // cold: the record is gathered from the rank-ordered image
            L.cpos[idx] = cold_pos_of_obs[i];
            L.cold_src[cold_pos_of_obs[i]] = ((L.tile[t].w >> 4) + (j - L.tile[t].z)) * WAVE + lane;
          }
        }
      }
    }
  };
  lpl_parallel(grid, n_threads, place_wg, cancel);
  lap("row placement (threads)");
  if (cancelled()) return;
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
  lap("partial records");
}

}  // namespace povar
