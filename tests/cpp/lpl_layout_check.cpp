// Host-only check of the lane-per-landmark layout builder (povar_amd/csrc/lpl_layout.hpp): reads a problem dumped
// by tests/test_lpl_layout.py, builds the layout for a given grid / LDS capacity, verifies its invariants and prints
// one JSON line of statistics.  No HIP runtime call is made (runs without a GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../povar_amd/csrc/lpl_layout.hpp"

using namespace povar;

template <class T>
static std::vector<T> read_vec(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<T> v(n / sizeof(T));
  if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) std::exit(2);
  std::fclose(f);
  return v;
}

#define CHECK(c)                                                      \
  do {                                                                \
    if (!(c)) { std::printf("FAILED %s line %d\n", #c, __LINE__); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const int n_cams = std::atoi(argv[1]), grid = std::atoi(argv[5]), n_acc = std::atoi(argv[6]);
  const auto lm_off = read_vec<int32_t>(argv[2]);
  const auto cam_idx = read_vec<int32_t>(argv[3]);
  const auto obs = read_vec<double>(argv[4]);
  const int n_lms = (int)lm_off.size() - 1;
  const int64_t n_obs = lm_off[n_lms];
  // popularity ranks (1-based, ties: lower index) as build_layout computes them
  std::vector<int64_t> cnt(n_cams, 0);
  for (int64_t i = 0; i < n_obs; ++i) cnt[cam_idx[i]]++;
  std::vector<int> order(n_cams), rank1(n_cams);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
  for (int r = 0; r < n_cams; ++r) rank1[order[r]] = r + 1;
  std::vector<int> slot_of_obs(n_obs);
  std::iota(slot_of_obs.begin(), slot_of_obs.end(), 0);
  LplLayout L;
  const bool place = std::getenv("LPL_CHECK_NOPLACE") == nullptr;  // the natural row order povar_create starts on
  build_lpl(n_cams, n_lms, lm_off.data(), cam_idx.data(), obs.data(), rank1, slot_of_obs, (size_t)n_obs, grid, n_acc, L, place);
  int same_but_rows = -1;
  if (std::getenv("LPL_CHECK_BOTH")) {
    // the other row order: everything but the six row-order arrays must be the same (povar_create swaps only those)
    LplLayout M;
    build_lpl(n_cams, n_lms, lm_off.data(), cam_idx.data(), obs.data(), rank1, slot_of_obs, (size_t)n_obs, grid, n_acc, M, !place);
    auto eq4 = [](const std::vector<int4>& a, const std::vector<int4>& b) {
      return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(int4)) == 0;
    };
    auto eq2 = [](const std::vector<int2>& a, const std::vector<int2>& b) {
      return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(int2)) == 0;
    };
    same_but_rows = eq4(L.tile, M.tile) && L.seg == M.seg && L.wg_tile_off == M.wg_tile_off && L.wg_cam_off == M.wg_cam_off &&
                    L.wg_cams == M.wg_cams && L.wg_slot_rec == M.wg_slot_rec && eq2(L.part_range, M.part_range) &&
                    L.cold_lm == M.cold_lm && eq2(L.cold_range, M.cold_range) && L.rows == M.rows &&
                    L.n_part_rec == M.n_part_rec && L.max_slots == M.max_slots && L.n_global == M.n_global &&
                    L.hubs == M.hubs && L.strategy == M.strategy && L.uv.size() == M.uv.size() && L.cold_rows == M.cold_rows &&
                    L.cold_src.size() == M.cold_src.size() &&
                    L.lm_of.size() == M.lm_of.size();
    CHECK(same_but_rows == 1);
  }
  // ---- invariants
  CHECK(L.max_slots <= n_acc);
  CHECK((int)L.wg_tile_off.size() == grid + 1 && (int)L.wg_cam_off.size() == grid + 1);
  CHECK(L.wg_tile_off[grid] == (int)L.tile.size());
  CHECK((int64_t)L.uv.size() == L.rows * 64 && L.cw.size() == L.uv.size() && L.cpos.size() == L.uv.size());
  // every observation appears exactly once, with its (u, v); resident slots point at its camera; cold ones at a
  // position of the cold view that belongs to its camera and names its landmark
  std::vector<char> seen(n_obs, 0);
  int64_t n_cold = 0, n_placed = 0, min_rows = 1 << 30;
  std::vector<int64_t> wg_rows(grid, 0), wg_obs(grid, 0);
  std::vector<int> lm_of(n_obs);
  for (int l = 0; l < n_lms; ++l)
    for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) lm_of[i] = l;
  for (int w = 0; w < grid; ++w) {
    const int c0 = L.wg_cam_off[w], nw = L.wg_cam_off[w + 1] - c0;
    for (int t = L.wg_tile_off[w]; t < L.wg_tile_off[w + 1]; ++t) {
      const int4 ti = L.tile[t];
      CHECK(ti.y >= 2 && ti.z <= ti.y);
      wg_rows[w] += ti.y;
      min_rows = std::min<int64_t>(min_rows, ti.y);
      if (t > L.wg_tile_off[w]) CHECK(L.tile[t - 1].y >= ti.y);  // longest first
      for (int j = 0; j < ti.y; ++j)
        for (int lane = 0; lane < 64; ++lane) {
          const size_t idx = ((size_t)ti.x + j) * 64 + lane;
          const int cw = L.cw[idx];
          if (j < ti.z) CHECK(cw >= 0);
          if (cw == -1) continue;
          ++n_placed;
          wg_obs[w]++;
          if (cw >= 0) {
            CHECK(lpl_cw_slot(cw) < nw && (cw >> 16) < 4);
          } else {
            ++n_cold;
            CHECK(L.cpos[idx] >= 0 && L.cpos[idx] < (int)L.cold_lm.size());
            // where the row kernels leave this observation's q (lpl_cold_q) and where the per-camera kernel looks for it
            const int64_t at = ((int64_t)(ti.w >> 4) + (j - ti.z)) * 64 + lane;
            CHECK(j >= ti.z && at >= 0 && at < L.cold_rows * 64 && L.cold_src[L.cpos[idx]] == (int)at);
          }
        }
    }
  }
  for (int64_t i = 0; i < n_obs; ++i) {
    const int idx = L.of_slot[i];
    CHECK(idx >= 0 && !seen[i]);
    seen[i] = 1;
    CHECK(L.uv[idx].x == obs[2 * i] && L.uv[idx].y == obs[2 * i + 1]);
    const int cw = L.cw[idx], r0 = rank1[cam_idx[i]] - 1;
    // which workgroup owns this row
    const int row = idx / 64;
    int w = 0;
    {
      int lo = 0, hi = grid;  // tiles are in workgroup order and rows ascend with the tile index
      while (lo + 1 < hi) {
        const int mid = (lo + hi) / 2;
        (L.wg_tile_off[mid] < (int)L.tile.size() && L.tile[L.wg_tile_off[mid]].x <= row && L.wg_tile_off[mid] < L.wg_tile_off[grid] ? lo : hi) = mid;
      }
      w = lo;
      while (w + 1 < grid && L.wg_tile_off[w + 1] < (int)L.tile.size() && L.tile[L.wg_tile_off[w + 1]].x <= row) ++w;
      while (w > 0 && (L.wg_tile_off[w] >= (int)L.tile.size() || L.tile[L.wg_tile_off[w]].x > row)) --w;
    }
    if (cw >= 0) {
      CHECK(L.wg_cams[L.wg_cam_off[w] + lpl_cw_slot(cw)] == r0);
    } else {
      CHECK(-2 - cw == r0);
      const int p = L.cpos[idx];
      CHECK(p >= L.cold_range[cam_idx[i]].x && p < L.cold_range[cam_idx[i]].y && L.cold_lm[p] == lm_of[i]);
    }
    // the landmark's lanes
    const int lp = L.lm_pos[lm_of[i]], pos = lp & ((1 << 26) - 1), lanes = ((lp >> 26) & 63) + 1;
    CHECK(idx % 64 >= (pos & 63) && idx % 64 < (pos & 63) + lanes);
  }
  CHECK(n_placed == n_obs && n_cold == (int64_t)L.cold_lm.size());
  // partial records: every slot of every workgroup has its own record inside its camera's run
  std::vector<char> rec_used(L.n_part_rec, 0);
  for (int w = 0; w < grid; ++w)
    for (int s = L.wg_cam_off[w]; s < L.wg_cam_off[w + 1]; ++s) {
      const int rec = L.wg_slot_rec[s], cam = order[L.wg_cams[s]];
      CHECK(rec >= L.part_range[cam].x && rec < L.part_range[cam].y && !rec_used[rec]);
      rec_used[rec] = 1;
    }
  for (int r = 0; r < L.n_part_rec; ++r) CHECK(rec_used[r]);
  // LDS collision statistics of the placement: extra lanes per accumulator bank per row half (each costs 8 LDS
  // cycles on each of the 12 ds_add_f64) and extra records per bank quad per ds_read_b128 lane group
  double extra_a = 0, extra_r = 0;
  {
    const int hubs = L.hubs;
    auto read_group = [](int lane) {
      const int l = lane & 31;
      const int g = (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1;
      return g + 2 * (lane >> 5);
    };
    for (int64_t r = 0; r < L.rows; ++r) {
      int ca[2][32] = {}, cr[4][16] = {}, sr[4][16];
      for (int g = 0; g < 4; ++g) for (int b = 0; b < 16; ++b) sr[g][b] = -1;
      for (int lane = 0; lane < 64; ++lane) {
        const int s = L.cw[r * 64 + lane];
        if (s < 0) continue;
        ca[lane >> 5][lpl_acc_slot(s, hubs) & 31]++;
        const int g = read_group(lane);
        const int ss = lpl_cw_slot(s);
        if (sr[g][ss & 15] != ss) { cr[g][ss & 15]++; sr[g][ss & 15] = ss; }  // same record next to itself: broadcast
      }
      for (int hlf = 0; hlf < 2; ++hlf) { int m = 1; for (int b = 0; b < 32; ++b) m = std::max(m, ca[hlf][b]); extra_a += m - 1; }
      for (int g = 0; g < 4; ++g) { int m = 1; for (int b = 0; b < 16; ++b) m = std::max(m, cr[g][b]); extra_r += m - 1; }
    }
    {  // by tile height (stderr): where the collisions sit
      double ea[16] = {}, er[16] = {}, nrow[16] = {};
      for (size_t t = 0; t < L.tile.size(); ++t) {
        const int R = std::min(L.tile[t].y, 15);
        for (int j = 0; j < L.tile[t].y; ++j) {
          const int64_t r = (int64_t)L.tile[t].x + j;
          int ca[2][32] = {}, cr[4][16] = {}, sr[4][16];
          for (int g = 0; g < 4; ++g) for (int b = 0; b < 16; ++b) sr[g][b] = -1;
          for (int lane = 0; lane < 64; ++lane) {
            const int sl = L.cw[r * 64 + lane];
            if (sl < 0) continue;
            ca[lane >> 5][lpl_acc_slot(sl, hubs) & 31]++;
            const int g = read_group(lane);
            const int ss = lpl_cw_slot(sl);
            if (sr[g][ss & 15] != ss) { cr[g][ss & 15]++; sr[g][ss & 15] = ss; }
          }
          for (int hlf = 0; hlf < 2; ++hlf) { int m = 1; for (int b = 0; b < 32; ++b) m = std::max(m, ca[hlf][b]); ea[R] += m - 1; }
          for (int g = 0; g < 4; ++g) { int m = 1; for (int b = 0; b < 16; ++b) m = std::max(m, cr[g][b]); er[R] += m - 1; }
          nrow[R] += 1;
        }
      }
      for (int R = 2; R < 16; ++R)
        if (nrow[R] > 0) std::fprintf(stderr, "tiles of %d rows: %.0f rows, extra atomics %.3f, extra reads %.3f\n", R, nrow[R], ea[R] / nrow[R] / 2, er[R] / nrow[R] / 4);
    }
    extra_a /= (double)L.rows * 2;
    extra_r /= (double)L.rows * 4;
    // lower bound of extra_a for the given tile membership and lane halves: a bank hit by deg observations in a
    // half-tile of R rows repeats at least ceil(deg / R) - 1 times in some row (printed to stderr)
    double bound = 0, bound_mean = 0, bm_R[32] = {}, rows_R[32] = {};
    for (size_t t = 0; t < L.tile.size(); ++t) {
      const int R = L.tile[t].y;
      for (int hlf = 0; hlf < 2; ++hlf) {
        int deg[32] = {};
        for (int j = 0; j < R; ++j)
          for (int lane = hlf * 32; lane < hlf * 32 + 32; ++lane) {
            const int sl = L.cw[((size_t)L.tile[t].x + j) * 64 + lane];
            if (sl >= 0) deg[lpl_acc_slot(sl, hubs) & 31]++;
          }
        int mx = 0;
        for (int b = 0; b < 32; ++b) mx = std::max(mx, deg[b]);
        // the best any placement can do: the heaviest bank spread evenly; averaged over the rows
        bound += (double)std::max(0, (mx + R - 1) / R - 1) ;
        bound_mean += std::max(0.0, (double)mx / R - 1.0);
        if (R < 32) { bm_R[R] += std::max(0.0, (double)mx / R - 1.0) * R; rows_R[R] += R; }
      }
    }
    std::fprintf(stderr, "lower bound of the MEAN extra atomic lanes per row half for this tile membership: %.3f\n", bound_mean / (L.tile.size() * 2));
    for (int R = 2; R < 32; ++R)
      if (rows_R[R] > 0) std::fprintf(stderr, "  tiles of %d rows: bound %.3f (row-weighted)\n", R, bm_R[R] / rows_R[R]);
    std::fprintf(stderr, "lower bound of the worst row per half-tile (max over rows, not the mean): %.3f\n", bound / (L.tile.size() * 2));
  }
  int64_t mx = 0, mn = 1LL << 60;
  for (int w = 0; w < grid; ++w) { mx = std::max(mx, wg_rows[w]); mn = std::min(mn, wg_rows[w]); }
  // fingerprint of everything the kernels read (the layout must not depend on the number of host threads)
  unsigned long long fp = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t bytes) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < bytes; ++i) { fp ^= b[i]; fp *= 1099511628211ull; }
  };
  mix(L.cw.data(), L.cw.size() * sizeof(int)); mix(L.cpos.data(), L.cpos.size() * sizeof(int));
  mix(L.uv.data(), L.uv.size() * sizeof(double2)); mix(L.lm_pos.data(), L.lm_pos.size() * sizeof(int));
  mix(L.lm_of.data(), L.lm_of.size() * sizeof(int)); mix(L.seg.data(), L.seg.size() * sizeof(int));
  mix(L.tile.data(), L.tile.size() * sizeof(int4)); mix(L.wg_cams.data(), L.wg_cams.size() * sizeof(int));
  mix(L.wg_slot_rec.data(), L.wg_slot_rec.size() * sizeof(int)); mix(L.cold_lm.data(), L.cold_lm.size() * sizeof(int));
  mix(L.of_slot.data(), L.of_slot.size() * sizeof(int));
  mix(L.cold_src.data(), L.cold_src.size() * sizeof(int));
  std::printf("{\"fingerprint\": \"%016llx\", ", fp);
  std::printf("\"strategy\": \"%s\", \"hubs\": %d, \"placed\": %d, \"same_but_rows\": %d, ", L.strategy ? "ranges" : "grid", L.hubs,
              place ? 1 : 0, same_but_rows);
  std::printf("\"ok\": 1, \"n_global\": %d, \"n_tail\": %d, \"grid\": [%d, %d], \"max_slots\": %d, \"rows\": %lld, "
              "\"tiles\": %zu, \"cold\": %lld, \"cold_frac\": %.5f, \"pad_frac\": %.5f, \"wg_rows_min\": %lld, "
              "\"wg_rows_max\": %lld, \"part_recs\": %d, \"extra_atomic_lanes_per_half\": %.3f, \"extra_records_per_read_group\": %.3f}\n",
              L.n_global, L.n_tail, L.grid_a, L.grid_b, L.max_slots, (long long)L.rows, L.tile.size(),
              (long long)n_cold, (double)n_cold / n_obs, (double)L.rows * 64 / n_obs - 1.0, (long long)mn, (long long)mx,
              L.n_part_rec, extra_a, extra_r);
  return 0;
}
