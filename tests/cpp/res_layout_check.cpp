// Host-only check of the layout builder of the resident power series (povar_amd/csrc/res_layout.hpp): reads a problem
// dumped by tests/test_res_layout.py, builds the layout, verifies the invariants series_res relies on and prints one JSON
// line of statistics.  No HIP runtime call (runs without a GPU).
// usage: res_layout_check n_cams lm_off.bin cam_idx.bin obs.bin W NW R HMIN HMAX LSMAX [force_order]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <set>
#include <vector>

#include "../../povar_amd/csrc/res_layout.hpp"

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
  if (argc < 11) return 2;
  const int n_cams = std::atoi(argv[1]), W = std::atoi(argv[5]), NW = std::atoi(argv[6]), RR = std::atoi(argv[7]),
            hmin = std::atoi(argv[8]), hmax = std::atoi(argv[9]), ls_max = std::atoi(argv[10]);
  const int force_order = argc > 11 ? std::atoi(argv[11]) : -1;
  const auto lm_off = read_vec<int32_t>(argv[2]);
  const auto cam_idx = read_vec<int32_t>(argv[3]);
  const auto obs = read_vec<double>(argv[4]);
  const int n_lms = (int)lm_off.size() - 1;
  const int64_t n_obs = lm_off[n_lms];
  std::vector<int64_t> cnt(n_cams, 0);
  for (int64_t i = 0; i < n_obs; ++i) cnt[cam_idx[i]]++;
  std::vector<int> order(n_cams), rank1(n_cams);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
  for (int r = 0; r < n_cams; ++r) rank1[order[r]] = r + 1;
  std::vector<int> slot_of_obs(n_obs);
  for (int64_t i = 0; i < n_obs; ++i) slot_of_obs[i] = (int)(n_obs - 1 - i);  // any bijection: the layout only carries it
  ResLayout R;
  const auto t0 = std::chrono::steady_clock::now();
  build_res(n_cams, n_lms, lm_off.data(), cam_idx.data(), obs.data(), rank1, slot_of_obs, W, NW, RR, hmin, hmax, ls_max, R, force_order);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!R.fits) {
    std::printf("{\"ok\": 1, \"fits\": 0, \"why\": \"%s\", \"max_lm\": %d, \"max_cam\": %d, \"build_ms\": %.1f}\n", R.why, R.max_lm,
                R.max_cam, ms);
    return 0;
  }
  const int T = NW * WAVE, H = R.H;
  CHECK(R.W >= 1 && R.W <= RES_MAX_WG && R.W <= W && R.NW == NW && R.R == RR && H >= hmin && H <= hmax && (H & (H - 1)) == 0);
  CHECK((int)R.lm_off.size() == R.W + 1 && R.lm_off[R.W] == n_lms && (int)R.lm_id.size() == n_lms);
  CHECK((int)R.cam_off.size() == R.W + 1 && (int)R.cam_id.size() == R.cam_off[R.W] && R.cam_zi.size() == R.cam_id.size() && R.n_rec == R.cam_off[R.W]);
  const size_t n_pos = (size_t)R.W * RR * T;
  CHECK(R.lane_cam.size() == n_pos && R.lane_seg.size() == n_pos);
  CHECK(R.uv.size() == n_pos * H && R.lslot.size() == R.uv.size() && R.oslot.size() == R.uv.size());
  CHECK(R.wave_h.size() == (size_t)R.W * RR * NW);
  CHECK(R.lds_bytes <= (size_t)RES_LDS_BYTES && R.LS >= 1 && R.LS <= ls_max && R.max_lm <= R.LS * T);
  // every landmark in exactly one slot, no empty workgroup
  std::vector<int> lm_wg(n_lms, -1);
  for (int g = 0; g < R.W; ++g) {
    CHECK(R.lm_off[g + 1] > R.lm_off[g]);
    for (int s = R.lm_off[g]; s < R.lm_off[g + 1]; ++s) {
      CHECK(R.lm_id[s] >= 0 && R.lm_id[s] < n_lms && lm_wg[R.lm_id[s]] < 0);
      lm_wg[R.lm_id[s]] = g;
    }
  }
  // observations by (landmark, camera) -> index, uv
  std::map<std::pair<int, int>, int64_t> where;
  for (int l = 0; l < n_lms; ++l)
    for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) where[{l, cam_idx[i]}] = i;
  std::vector<char> seen(n_obs, 0);
  int64_t lanes_used = 0, rows_issued = 0, single_run = 0;
  for (int g = 0; g < R.W; ++g) {
    const int nC = R.cam_off[g + 1] - R.cam_off[g];
    std::set<int> cams_of_wg;
    for (int s = 0; s < nC; ++s) {
      const int c = R.cam_id[R.cam_off[g] + s];
      CHECK(c >= 0 && c < n_cams && cams_of_wg.insert(c).second && R.cam_zi[R.cam_off[g] + s] == rank1[c] - 1);
      if (s > 0) CHECK(rank1[c] > rank1[R.cam_id[R.cam_off[g] + s - 1]]);  // most observed first
    }
    std::map<int, int> runs_of_slot, single_flag;
    for (int r = 0; r < RR; ++r)
      for (int wv = 0; wv < NW; ++wv) {
        const int wh = R.wave_h[((size_t)g * RR + r) * NW + wv], hrows = wh & 255, dup = (wh >> 8) & 1, steps = (wh >> 12) & 15;
        CHECK(hrows <= H && steps >= 1 && steps <= 4);
        rows_issued += hrows;
        bool any_dup = false;
        for (int l = 0; l < WAVE; ++l) {
          const size_t t = (size_t)wv * WAVE + l, lane = ((size_t)g * RR + r) * T + t;
          const int ci = R.lane_cam[lane];
          if (ci < 0) {
            for (int j = 0; j < H; ++j) CHECK(R.lslot[(((size_t)g * RR + r) * H + j) * T + t] < 0);
            continue;
          }
          ++lanes_used;
          CHECK(ci < nC);
          const int cam = R.cam_id[R.cam_off[g] + ci];
          const int s0 = R.lane_seg[lane] & 255, s1 = (R.lane_seg[lane] >> 8) & 255;
          CHECK(s0 <= l && l <= s1 && s1 < WAVE);
          const size_t base = ((size_t)g * RR + r) * T + (size_t)wv * WAVE;
          for (int q = s0; q <= s1; ++q) CHECK(R.lane_cam[base + q] == ci && R.lane_seg[base + q] == R.lane_seg[lane]);
          single_flag[ci] = (R.lane_seg[lane] >> 16) & 1;
          if (s0 > 0) CHECK(R.lane_cam[base + s0 - 1] != ci);
          if (s1 + 1 < WAVE) CHECK(R.lane_cam[base + s1 + 1] != ci);
          if (l == s0) runs_of_slot[ci]++;
          if (s1 > s0) { any_dup = true; CHECK((1 << steps) >= std::min(s1 - s0 + 1, 16)); }
          bool ended = false;
          for (int j = 0; j < H; ++j) {
            const size_t row = (((size_t)g * RR + r) * H + j) * T + t;
            if (R.lslot[row] < 0) { ended = true; continue; }
            CHECK(!ended && j < hrows && R.lslot[row] % 3 == 0);
            const int slot = R.lslot[row] / 3;
            CHECK(slot < R.lm_off[g + 1] - R.lm_off[g]);
            const int lm = R.lm_id[R.lm_off[g] + slot];
            auto it = where.find({lm, cam});
            CHECK(it != where.end());
            const int64_t i = it->second;
            CHECK(!seen[i]);
            seen[i] = 1;
            CHECK(R.uv[row].x == obs[2 * i] && R.uv[row].y == obs[2 * i + 1] && R.oslot[row] == slot_of_obs[i]);
          }
        }
        CHECK(any_dup == (dup != 0));
      }
    CHECK((int)runs_of_slot.size() == nC);  // every camera slot of the workgroup has a chunk
    for (auto& kv : runs_of_slot) {
      single_run += kv.second == 1;
      CHECK(single_flag[kv.first] == (kv.second == 1 ? 1 : 0));
    }
  }
  for (int64_t i = 0; i < n_obs; ++i) CHECK(seen[i]);
  // owners: every camera once; its records = the slots that name it, each once, in workgroup order
  CHECK((int)R.own_off.size() == R.W + 1 && R.own_off[R.W] == n_cams && (int)R.own_cam.size() == n_cams);
  CHECK((int)R.oq_off.size() == R.W + 1 && R.oq_off[R.W] == R.n_rec && (int)R.oq_rec.size() == R.n_rec);
  CHECK(R.own_q.size() == R.own_cam.size() && R.own_zi.size() == R.own_cam.size());
  std::vector<int> owned(n_cams, 0), rec_seen(R.n_rec, 0);
  int64_t max_own_rec = 0;
  for (int g = 0; g < R.W; ++g) {
    int q = 0;
    for (int o = R.own_off[g]; o < R.own_off[g + 1]; ++o) {
      const int c = R.own_cam[o];
      CHECK(c >= 0 && c < n_cams && owned[c]++ == 0 && R.own_zi[o] == rank1[c] - 1);
      CHECK(R.own_q[o].x == q && R.own_q[o].y >= q);
      int prev = -1;
      for (q = R.own_q[o].x; q < R.own_q[o].y; ++q) {
        const int rec = R.oq_rec[(size_t)R.oq_off[g] + q];
        CHECK(rec >= 0 && rec < R.n_rec && rec_seen[rec]++ == 0 && R.cam_id[rec] == c && rec > prev);
        prev = rec;
      }
    }
    CHECK(q == R.oq_off[g + 1] - R.oq_off[g]);
    max_own_rec = std::max<int64_t>(max_own_rec, q);
  }
  for (int c = 0; c < n_cams; ++c) CHECK(owned[c] == 1);
  for (int r = 0; r < R.n_rec; ++r) CHECK(rec_seen[r] == 1);
  std::printf("{\"ok\": 1, \"fits\": 1, \"W\": %d, \"H\": %d, \"LS\": %d, \"order\": %d, \"n_rec\": %d, \"single_run\": %lld, \"max_lm\": %d, "
              "\"max_cam\": %d, \"max_oq\": %d, \"max_own\": %d, \"max_chunks\": %d, \"max_own_rec\": %lld, \"lanes_used\": %lld, "
              "\"lane_fill\": %.3f, \"row_fill\": %.3f, \"lds_bytes\": %zu, \"build_ms\": %.1f}\n",
              R.W, H, R.LS, R.order, R.n_rec, (long long)single_run, R.max_lm, R.max_cam, R.max_oq, R.max_own, R.max_chunks,
              (long long)max_own_rec, (long long)lanes_used, (double)lanes_used / ((double)R.W * RR * T),
              (double)n_obs / std::max<double>(1.0, (double)rows_issued * WAVE), R.lds_bytes, ms);
  return 0;
}
