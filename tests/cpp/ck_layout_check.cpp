// Host-only check of the camera-chunk layout builder (povar_amd/csrc/ck_layout.hpp): reads a problem dumped by
// tests/test_ck_layout.py, builds the lane-per-landmark layout and the camera-chunk layout derived from it, verifies
// the invariants e0_ck relies on and prints one JSON line of statistics.  No HIP runtime call (runs without a GPU).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../povar_amd/csrc/ck_layout.hpp"

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
  if (argc < 8) return 2;
  const int n_cams = std::atoi(argv[1]), grid = std::atoi(argv[5]), n_acc = std::atoi(argv[6]), n_waves = std::atoi(argv[7]);
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
  std::iota(slot_of_obs.begin(), slot_of_obs.end(), 0);
  LplLayout L;
  build_lpl(n_cams, n_lms, lm_off.data(), cam_idx.data(), obs.data(), rank1, slot_of_obs, (size_t)n_obs, grid, n_acc, L,
            std::getenv("LPL_CHECK_NOPLACE") == nullptr);
  CkLayout K;
  const auto t0 = std::chrono::steady_clock::now();
  const int ng = std::getenv("CK_CHECK_NG") ? std::atoi(std::getenv("CK_CHECK_NG")) : 1;
  const bool step2 = std::getenv("CK_CHECK_STEP2") != nullptr;  // the shape of e0_ck_h's layout
  // step 1 as shipped (cold lanes leave q in the parent's cold view: CkLayout::cpos), or CK_CHECK_COLD_RECORDS=1: a record per cold chunk
  const CkShape shape = step2 ? ck_shape_step2() : std::getenv("CK_CHECK_COLD_RECORDS") ? CkShape() : ck_shape_step1();
  build_ck(L, n_cams, grid, order, n_waves, K, std::getenv("CK_CHECK_NOPLACE") == nullptr, CK_HMAX, ng, shape);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // ---- invariants
  CHECK((int)K.bt_off.size() == grid * K.nb + 1 && K.bt_off.back() == (int)K.tile.size());
  // image points: 16-byte doubles, or packed (two int32 of micro-units) when every one of them is a six-decimal number
  // (CK_CHECK_WANT_PACKED=0|1: what the caller expects of this problem); either way uv_at() must return the problem's bits
  CHECK(K.packed ? (K.uv.empty() && K.uvp.size() == K.n_uv) : (K.uvp.empty() && K.uv.size() == K.n_uv));
  if (const char* e = std::getenv("CK_CHECK_WANT_PACKED")) CHECK(K.packed == (e[0] == '1'));
  if (step2) CHECK(!K.packed);
  auto uv_at = [&](size_t idx) { return K.packed ? make_double2(ck_unpack_uv_host(K.uvp[idx].x), ck_unpack_uv_host(K.uvp[idx].y)) : K.uv[idx]; };
  {  // the pack / unpack pair by itself: six-decimal numbers come back bit for bit, anything else is refused
    int32_t k = 0;
    const double good[] = {0.0, 1e-6, -1e-6, 332.65, -0.000001, 2147.483646, -2147.483646, 123.456789, 1999.999999};
    for (double x : good) { CHECK(ck_pack_one(x, k)); const double y = ck_unpack_uv_host(k); CHECK(std::memcmp(&x, &y, 8) == 0 || x == 0.0); }
    const double bad[] = {1e-7, 0.1234567, 2147.483648, -3000.0, 1.0 / 3.0, 332.65 + 1e-12};
    for (double x : bad) CHECK(!ck_pack_one(x, k));
    uint64_t state = 88172645463325252ull;
    for (int i = 0; i < 200000; ++i) {  // k / 10^6 for random k: what strtod makes of a "%.6f" string
      state ^= state << 13; state ^= state >> 7; state ^= state << 17;
      const int32_t kk = (int32_t)(state >> 33) - (1 << 30);
      const double x = (double)kk / 1e6;
      CHECK(ck_pack_one(x, k) && k == kk);
      const double y = ck_unpack_uv_host(k);
      CHECK(std::memcmp(&x, &y, 8) == 0);
    }
  }
  CHECK((int64_t)K.n_uv == (K.rows + CK_HMAX) * 64 && K.src.size() == K.n_uv && (int64_t)K.li.size() == (K.li_rows + CK_HMAX) * 64);
  // (a fixed-stride shape may have taken its wide stride: K.stride; the accumulators are then capped by what fits beside it)
  CkShape shape_k = shape;
  if (shape.max_slots != INT_MAX) {
    CHECK(K.stride == shape.max_slots || (shape.wide_slots > 0 && K.stride == shape.wide_slots));
    shape_k.max_slots = K.stride;
  } else {
    CHECK(K.stride == 0);
  }
  CHECK(ck_lds_bytes_shape(shape_k, K.slots, K.max_acc, K.ng) <= (size_t)CK_LDS_BYTES && K.nb % K.ng == 0 && K.slots <= shape_k.max_slots);
  if (const char* e = std::getenv("CK_CHECK_WANT_STRIDE")) CHECK(K.stride == std::atoi(e));
  int64_t n_capped = 0;
  std::vector<int> lm_of_obs(n_obs);
  for (int l = 0; l < n_lms; ++l)
    for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) lm_of_obs[i] = l;
  // observation of every lane-per-landmark row slot
  std::vector<int> obs_of_slot(L.uv.size(), -1);
  for (int64_t i = 0; i < n_obs; ++i) obs_of_slot[L.of_slot[i]] = (int)i;
  std::vector<char> seen(n_obs, 0), rec_used(K.n_part_rec, 0), cold_seen(L.cold_lm.size(), 0);
  int64_t n_cold_q = 0;
  CHECK((!K.cold_q || shape.cold_q) && (K.cold_q ? K.cpos.size() == K.n_uv : K.cpos.empty()));  // (cold_q only up to 8 % cold observations)
  int64_t n_placed = 0, hist[CK_HMAX + 1] = {};
  for (int w = 0; w < grid; ++w) {
    const int n_par = L.wg_cam_off[w + 1] - L.wg_cam_off[w];
    const int nw = std::min(n_par, K.max_acc);  // accumulator slots of this workgroup (all of the parent's unless capped)
    std::vector<int> rank_of_acc(nw, -1);      // every accumulator serves ONE camera, a camera with one has no cold chunk here
    std::vector<char> cold_here(n_cams, 0);
    const int t0w = L.wg_tile_off[w];
    for (int b = 0; b < K.nb; ++b)
      for (int t = K.bt_off[(size_t)w * K.nb + b]; t < K.bt_off[(size_t)w * K.nb + b + 1]; ++t) {
        const int4 ti = K.tile[t];
        CHECK(ti.y >= 1 && ti.y <= CK_HMAX);
        hist[ti.y]++;
        if (t > K.bt_off[(size_t)w * K.nb + b]) CHECK(K.tile[t - 1].y >= ti.y);  // longest first inside a batch
        for (int lane = 0; lane < 64; ++lane) {
          const int rank = K.lane_cam[(size_t)t * 64 + lane], acc = K.lane_acc[(size_t)t * 64 + lane];
          const int sg = K.lane_seg[(size_t)t * 64 + lane], s_first = sg & 255, s_last = sg >> 8;
          CHECK(s_first <= lane && lane <= s_last && s_last < 64);
          int n_lane = 0;
          for (int j = 0; j < ti.y; ++j) {
            const size_t idx = ((size_t)ti.x + j) * 64 + lane;
            const uint32_t word = K.li[((size_t)ti.w + (j >> 1)) * 64 + lane];
            const uint32_t li3 = (j & 1) ? word >> 16 : word & 0xffffu;  // 3 x slot
            const int s = K.src[idx];
            if (s < 0) { CHECK(li3 == CK_NONE); continue; }
            CHECK(li3 % (uint32_t)K.li_mul == 0);
            const uint32_t li = li3 / (uint32_t)K.li_mul;
            ++n_lane;
            CHECK(rank >= 0 && (int)li < K.slots);
            const int i = obs_of_slot[s];
            CHECK(i >= 0 && !seen[i]);
            seen[i] = 1;
            ++n_placed;
            const double2 uvi = uv_at(idx);
            CHECK(std::memcmp(&uvi.x, &obs[2 * (size_t)i], 8) == 0 && std::memcmp(&uvi.y, &obs[2 * (size_t)i + 1], 8) == 0);
            CHECK(rank1[cam_idx[i]] - 1 == rank);
            // the landmark slot: tile (slot / 64) of this batch, a lane of the landmark
            const int lt = t0w + b + K.nb * (int)(li / 64);
            CHECK(lt < L.wg_tile_off[w + 1] && L.lm_of[(size_t)lt * 64 + (li & 63)] == lm_of_obs[i]);
            CHECK((L.seg[(size_t)lt * 64 + (li & 63)] & 255) == (int)(li & 63));
          }
          if (rank < 0) { CHECK(n_lane == 0); continue; }
          CHECK(n_lane >= 1);
          if (acc >= 0) {
            CHECK(acc < nw && (rank_of_acc[acc] < 0 || rank_of_acc[acc] == rank));
            rank_of_acc[acc] = rank;
            if (nw == n_par) CHECK(L.wg_cams[L.wg_cam_off[w] + acc] == rank);
            {  // the record the accumulator is flushed to is one of this camera's
              const int rec = K.slot_rec[(size_t)L.wg_cam_off[w] + acc], cam = order[rank];
              CHECK(rec >= K.part_range[cam].x && rec < K.part_range[cam].y);
            }
            for (int x = s_first; x <= s_last; ++x) CHECK(K.lane_acc[(size_t)t * 64 + x] == acc);
            if (s_first != s_last) CHECK(ti.z & CK_FLAG_DUP);
          } else if (K.cold_q) {
            // no record: every observation of the lane names its own place in the parent's cold view, inside its camera's run
            CHECK(ti.z & CK_FLAG_COLD);
            CHECK(acc == -1 && s_first == lane && s_last == lane);
            const int cam = order[rank];
            for (int j = 0; j < ti.y; ++j) {
              const size_t idx = ((size_t)ti.x + j) * 64 + lane;
              if (K.src[idx] < 0) continue;
              const int cp = K.cpos[idx];
              CHECK(cp == L.cpos[K.src[idx]] && cp >= L.cold_range[cam].x && cp < L.cold_range[cam].y && !cold_seen[cp]);
              cold_seen[cp] = 1;
              ++n_cold_q;
            }
          } else {
            CHECK(ti.z & CK_FLAG_COLD);
            cold_here[rank] = 1;
            for (int j = 0; j < ti.y; ++j) {  // (observations of a camera with a slot in the parent layout but none here)
              const size_t idx = ((size_t)ti.x + j) * 64 + lane;
              if (K.src[idx] >= 0 && L.cw[K.src[idx]] >= 0) ++n_capped;
            }
            const int rec = ~acc, cam = order[rank];
            CHECK(rec >= K.part_range[cam].x && rec < K.part_range[cam].y && !rec_used[rec]);
            rec_used[rec] = 1;
            CHECK(s_first == lane && s_last == lane);
          }
          if (acc >= 0 && K.cold_q)
            for (int j = 0; j < ti.y; ++j) CHECK(K.cpos[((size_t)ti.x + j) * 64 + lane] == -1);
        }
      }
    for (int a = 0; a < nw; ++a) {
      const int rec = K.slot_rec[(size_t)L.wg_cam_off[w] + a];
      CHECK(rec >= 0 && rec < K.n_part_rec && !rec_used[rec]);
      rec_used[rec] = 1;
      if (nw == n_par) {
        const int cam = order[L.wg_cams[L.wg_cam_off[w] + a]];
        CHECK(rec >= K.part_range[cam].x && rec < K.part_range[cam].y);
      }
      if (rank_of_acc[a] >= 0) CHECK(!cold_here[rank_of_acc[a]]);
    }
    for (int a = 0; a < nw; ++a)
      for (int c = a + 1; c < nw; ++c) CHECK(rank_of_acc[a] < 0 || rank_of_acc[a] != rank_of_acc[c]);
  }
  CHECK(n_placed == n_obs && n_capped == K.n_capped_obs);
  for (int r = 0; r < K.n_part_rec; ++r) CHECK(rec_used[r]);
  if (K.cold_q) {  // every cold observation of the parent layout is written by exactly one lane; the records are the slots' alone
    CHECK(n_cold_q == (int64_t)L.cold_lm.size() && K.n_part_rec == (int)L.wg_cams.size());
    for (char c : cold_seen) CHECK(c);
  }
  // ---- what the bit-reproducible kernel (e0_ck_det) reads on top: observation counts per landmark lane, tickets per run total
  CHECK(K.lcnt_log2.size() == L.tile.size() * 64 && K.tick.size() == K.tile.size() * 64);
  {
    std::vector<int> cnt(L.tile.size() * 64, 0);
    for (int w = 0; w < grid; ++w)
      for (int b = 0; b < K.nb; ++b)
        for (int t = K.bt_off[(size_t)w * K.nb + b]; t < K.bt_off[(size_t)w * K.nb + b + 1]; ++t)
          for (int j = 0; j < K.tile[t].y; ++j)
            for (int lane = 0; lane < 64; ++lane) {
              const uint32_t word = K.li[((size_t)K.tile[t].w + (j >> 1)) * 64 + lane];
              const uint32_t li3 = (j & 1) ? word >> 16 : word & 0xffffu;
              if (li3 == CK_NONE) continue;
              const uint32_t li = li3 / (uint32_t)K.li_mul;
              cnt[(size_t)(L.wg_tile_off[w] + b + K.nb * (int)(li / 64)) * 64 + (li & 63)]++;
            }
    for (size_t i = 0; i < cnt.size(); ++i) {
      if (cnt[i] == 0) { CHECK(K.lcnt_log2[i] == 255); continue; }
      CHECK((1 << K.lcnt_log2[i]) >= cnt[i] && (K.lcnt_log2[i] == 0 || (1 << (K.lcnt_log2[i] - 1)) < cnt[i]));
    }
    // tickets: per (workgroup, slot) 0, 1, 2, ... in the order batches -> rounds of n_waves tiles -> tiles of a round from the
    // last to the first -> lanes; exactly the last lanes of the runs with a slot carry one
    for (int w = 0; w < grid; ++w) {
      std::vector<int> next(std::max(1, L.wg_cam_off[w + 1] - L.wg_cam_off[w]), 0);
      for (int b = 0; b < K.nb; ++b) {
        const int tb0 = K.bt_off[(size_t)w * K.nb + b], tb1 = K.bt_off[(size_t)w * K.nb + b + 1];
        for (int q0 = tb0; q0 < tb1; q0 += n_waves)
          for (int t = std::min(q0 + n_waves, tb1) - 1; t >= q0; --t)
            for (int lane = 0; lane < 64; ++lane) {
              const size_t i = (size_t)t * 64 + lane;
              const bool last = K.lane_cam[i] >= 0 && K.lane_acc[i] >= 0 && lane == (K.lane_seg[i] >> 8);
              if (last) CHECK(K.tick[i] == next[K.lane_acc[i]]++ && K.tick[i] < 65535);
              else CHECK(K.tick[i] == 0);
            }
      }
    }
  }
  for (int h = 1; h <= CK_HMAX; ++h)
    if (hist[h]) std::fprintf(stderr, "tiles of %2d rows: %lld\n", h, (long long)hist[h]);
  std::printf("{\"ok\": 1, \"nb\": %d, \"slots\": %d, \"tiles\": %zu, \"rows\": %lld, \"chunks\": %lld, \"cold_chunks\": %lld, "
              "\"obs_per_chunk\": %.3f, \"pad_frac\": %.4f, \"max_tiles_bt\": %d, \"part_rec\": %d, \"lpl_part_rec\": %d, "
              "\"extra_lanes_per_half_row\": %.4f, \"lds_bytes\": %zu, \"build_ms\": %.1f, \"lpl_rows\": %lld, \"packed\": %d, \"cold_q\": %d, \"cold_obs\": %zu, \"stride\": %d, \"max_acc\": %d, \"capped_obs\": %lld}\n",
              K.nb, K.slots, K.tile.size(), (long long)K.rows, (long long)K.n_chunks, (long long)K.n_cold_chunks,
              (double)n_obs / std::max<int64_t>(K.n_chunks, 1), 1.0 - (double)n_obs / ((double)K.rows * 64), K.max_tiles_bt,
              K.n_part_rec, L.n_part_rec, K.extra_lanes / (2.0 * std::max<int64_t>(K.rows, 1)), ck_lds_bytes_shape(shape_k, K.slots, K.max_acc, K.ng), ms,
              (long long)L.rows, K.packed ? 1 : 0, K.cold_q ? 1 : 0, L.cold_lm.size(), K.stride, K.max_acc, (long long)K.n_capped_obs);
  return 0;
}
