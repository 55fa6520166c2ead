// Per-tile statistics of the camera-chunk layout of a problem dumped as tests/test_ck_layout.py does (lm_off.bin, cam_idx.bin, obs.bin):
// rows, observations, distinct cameras and the cache lines a gather instruction touches under three record layouts, averaged over the
// workgroups per (batch, tile index).  hipcc -O2 -std=c++17 --offload-arch=gfx950 -o build/ck_tile_stats tools/ck_tile_stats.cpp;
// build/ck_tile_stats <n_cams> lm_off.bin cam_idx.bin obs.bin 256 520 16   (profiles/r06_ck_tile_stats_venice.txt)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <set>
#include <vector>
#include "../povar_amd/csrc/ck_layout.hpp"
using namespace povar;
template <class T> static std::vector<T> read_vec(const char* path) {
  FILE* f = std::fopen(path, "rb"); std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<T> v(n / sizeof(T)); if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) std::exit(2); std::fclose(f); return v;
}
int main(int argc, char** argv) {
  const int n_cams = std::atoi(argv[1]), grid = std::atoi(argv[5]), n_acc = std::atoi(argv[6]), n_waves = std::atoi(argv[7]);
  const auto lm_off = read_vec<int32_t>(argv[2]); const auto cam_idx = read_vec<int32_t>(argv[3]); const auto obs = read_vec<double>(argv[4]);
  const int n_lms = (int)lm_off.size() - 1; const int64_t n_obs = lm_off[n_lms];
  std::vector<int64_t> cnt(n_cams, 0); for (int64_t i = 0; i < n_obs; ++i) cnt[cam_idx[i]]++;
  std::vector<int> order(n_cams), rank1(n_cams); std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
  for (int r = 0; r < n_cams; ++r) rank1[order[r]] = r + 1;
  std::vector<int> slot_of_obs(n_obs); std::iota(slot_of_obs.begin(), slot_of_obs.end(), 0);
  LplLayout L;
  build_lpl(n_cams, n_lms, lm_off.data(), cam_idx.data(), obs.data(), rank1, slot_of_obs, (size_t)n_obs, grid, n_acc, L, false);
  CkLayout K;
  build_ck(L, n_cams, grid, order, n_waves, K, true, CK_HMAX, 1, ck_shape_step1());
  // averages over workgroups, per (batch, tile index)
  const int NT = 24;
  std::vector<double> rows(K.nb * NT, 0), fill(K.nb * NT, 0), cams(K.nb * NT, 0), extra(K.nb * NT, 0), same(K.nb * NT, 0), n(K.nb * NT, 0), cold(K.nb*NT,0), l8(K.nb*NT,0), l4(K.nb*NT,0), lz(K.nb*NT,0);
  for (int w = 0; w < grid; ++w)
    for (int b = 0; b < K.nb; ++b) {
      const int tb0 = K.bt_off[(size_t)w * K.nb + b], tb1 = K.bt_off[(size_t)w * K.nb + b + 1];
      for (int t = tb0; t < tb1 && t - tb0 < NT; ++t) {
        const int4 ti = K.tile[t]; const int k = b * NT + (t - tb0);
        std::set<int> cs, cs8, cs4, csz; int obs_n = 0; double ex = 0, sm = 0;
        for (int lane = 0; lane < 64; ++lane) if (K.lane_cam[(size_t)t * 64 + lane] >= 0) { const int r = K.lane_cam[(size_t)t * 64 + lane]; cs.insert(r); cs8.insert(r >> 3); cs4.insert(r >> 2); csz.insert((r * 96) >> 7); csz.insert((r * 96 + 95) >> 7); }
        for (int j = 0; j < ti.y; ++j) {
          int occ[2][32] = {}; std::set<uint32_t> slots; int nl = 0;
          for (int lane = 0; lane < 64; ++lane) {
            const uint32_t word = K.li[((size_t)ti.w + (j >> 1)) * 64 + lane];
            const uint32_t li3 = (j & 1) ? word >> 16 : word & 0xffffu;
            if (li3 == CK_NONE) continue;
            const uint32_t li = li3 / 3; ++obs_n; ++nl; occ[lane >> 5][li & 31]++; slots.insert(li);
          }
          for (int hf = 0; hf < 2; ++hf) { int mx = 1; for (int q = 0; q < 32; ++q) mx = std::max(mx, occ[hf][q]); ex += mx - 1; }
          sm += nl - (int)slots.size();
        }
        rows[k] += ti.y; fill[k] += obs_n; cams[k] += cs.size(); l8[k] += cs8.size(); l4[k] += cs4.size(); lz[k] += csz.size(); extra[k] += ex; same[k] += sm; n[k] += 1; cold[k] += (ti.z & 2) ? 1 : 0;
      }
    }
  for (int b = 0; b < K.nb; ++b) {
    std::printf("batch %d: tile, workgroups that have it, rows, observations, distinct cameras, cache lines touched by: the 96-byte z rows (all six loads) / ONE load of a piece-major image of 16-byte pieces / of 32-byte pieces\n", b);
    for (int i = 0; i < NT; ++i) { const int k = b * NT + i; if (n[k] == 0) continue;
      std::printf("   %2d  %5.0f  %5.1f  %6.1f  cams %6.1f  lines: z rows of 96 B %6.1f  16-byte pieces by rank %6.1f  32-byte %6.1f\n", i, n[k], rows[k] / n[k], fill[k] / n[k], cams[k] / n[k], lz[k]/n[k], l8[k] / n[k], l4[k] / n[k]); }
  }
  return 0;
}
