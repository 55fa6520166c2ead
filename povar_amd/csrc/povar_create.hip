// povar_create.hip -- layout construction and upload, povar_create / povar_destroy, read-back helpers, timings, layout info.
#include "povar_ctx.hpp"

std::mutex g_capture_mu;  // see povar_ctx::placer_cancel

// ------------------------------------------------------------------------------------------
// layout construction (host)
// ------------------------------------------------------------------------------------------
// number of cameras whose Jp^T s is accumulated in LDS (POVAR_HOT_ACC=<n> lowers it: tuning knob)
int hot_acc_cap(int n_cams) {
  int cap = HOT_ACC_MAX;
  if (const char* e = std::getenv("POVAR_HOT_ACC")) cap = std::max(1, std::min(HOT_ACC_MAX, std::atoi(e)));
  return std::min(n_cams, cap);
}

struct Layout {
  std::vector<double2> uv, cm_uv;
  std::vector<int> cam, lm, meta, hot_cams, cam_hot, cc_slot, cc_lm, cc_item_off, cc_cam_item_off, long_lm, long_first, long_cnt, cm_slot, cm_lm, item_off,
      item_cam, cam_item_off, slot_of_obs, cold_pos;
  // cold view "A" of the default mode when the problem has long landmarks: their observations of LDS-accumulated
  // cameras are accumulated inside e0_lm_cached too, so they are not cold (empty when there is no long landmark)
  std::vector<int> c2_lm, c2_pos;
  std::vector<int2> c2_range;
  int n_bins = 0;
};

// Temporaries that part A of the layout construction hands to part B.
struct LayoutTmp {
  std::vector<int> seg_first, seg_last, rank;
  std::vector<char> is_long;
  std::vector<int64_t> cnt;  // prefix sums of the observations per camera
};

// Part A: what everything else needs first -- the wave-bin slot of every observation, the popularity rank of every
// camera (the order of the record image, the key of the lane-per-landmark layout) and the camera-major work items.
void build_layout_a(int n_cams, int n_lms, const int32_t* lm_off, const int32_t* cam_idx, Layout& L, LayoutTmp& T) {
  const int64_t n_obs = lm_off[n_lms];
  L.slot_of_obs.resize(n_obs);
  // pass 1: assign slots.  Regular landmarks are packed greedily, in order, into 64-lane wave
  // bins that never split a landmark; a landmark with more than 64 observations gets
  // ceil(k/64) bins of its own and is handled by the lm_long driver.
  int bin = 0, fill = 0;
  T.seg_first.resize(n_obs);
  T.seg_last.resize(n_obs);
  T.is_long.assign(n_obs, 0);
  for (int l = 0; l < n_lms; ++l) {
    const int b = lm_off[l], k = lm_off[l + 1] - b;
    if (k == 0) continue;
    if (k > WAVE) {
      if (fill > 0) { ++bin; fill = 0; }
      L.long_lm.push_back(l);
      L.long_first.push_back(bin * WAVE);
      L.long_cnt.push_back(k);
      for (int j = 0; j < k; ++j) {
        L.slot_of_obs[b + j] = bin * WAVE + j;
        T.is_long[b + j] = 1;
      }
      bin += (k + WAVE - 1) / WAVE;
      continue;
    }
    if (fill + k > WAVE) { ++bin; fill = 0; }
    for (int j = 0; j < k; ++j) {
      L.slot_of_obs[b + j] = bin * WAVE + fill + j;
      T.seg_first[b + j] = fill;
      T.seg_last[b + j] = fill + k - 1;
    }
    fill += k;
  }
  if (fill > 0) ++bin;
  L.n_bins = std::max(bin, 1);
  std::vector<int64_t>& cnt = T.cnt;
  cnt.assign(n_cams + 1, 0);
  for (int64_t i = 0; i < n_obs; ++i) cnt[cam_idx[i] + 1]++;
  for (int c = 0; c < n_cams; ++c) cnt[c + 1] += cnt[c];
  // popularity rank of EVERY camera (1-based; ties: lower index): the record image (Dp::hot_rec) is in this order, so
  // the first n records are the LDS image of a kernel that caches n cameras and colder cameras gather theirs by rank
  std::vector<int> order(n_cams);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int a, int b) { return cnt[a + 1] - cnt[a] > cnt[b + 1] - cnt[b]; });
  T.rank.assign(n_cams, 0);
  L.hot_cams.assign(order.begin(), order.end());
  for (int r = 0; r < n_cams; ++r) T.rank[order[r]] = r + 1;
  L.cam_hot = T.rank;
  // camera-major work items of at most CM_ITEM_MAX observations of one camera
  L.cam_item_off.assign(n_cams + 1, 0);
  for (int c = 0; c < n_cams; ++c) {
    L.cam_item_off[c] = (int)L.item_cam.size();
    for (int64_t p = cnt[c]; p < cnt[c + 1]; p += CM_ITEM_MAX) {
      L.item_off.push_back((int)p);
      L.item_cam.push_back(c);
    }
  }
  L.cam_item_off[n_cams] = (int)L.item_cam.size();
  L.item_off.push_back((int)n_obs);
}

// Part B: the arrays of the lane-per-observation kernels (wave-bin slots, camera-major inverse index, cold views).
// Independent of the lane-per-landmark layout: povar_create builds the two side by side.
void build_layout_b(int n_cams, int n_lms, const int32_t* lm_off, const int32_t* cam_idx, const double* obs, Layout& L,
                    const LayoutTmp& T) {
  const int64_t n_obs = lm_off[n_lms];
  const std::vector<int>&seg_first = T.seg_first, &seg_last = T.seg_last, &rank = T.rank;
  const std::vector<char>& is_long = T.is_long;
  const std::vector<int64_t>& cnt = T.cnt;
  const size_t n_slots = (size_t)L.n_bins * WAVE;
  L.uv.assign(n_slots, make_double2(0, 0));
  L.cam.assign(n_slots, -1);
  L.lm.assign(n_slots, 0);
  L.meta.resize(n_slots);
  for (size_t s = 0; s < n_slots; ++s) {
    const int lane = (int)(s & 63);
    L.meta[s] = lane | (lane << 8);
  }
  for (int l = 0; l < n_lms; ++l)
    for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
      const int s = L.slot_of_obs[i];
      L.uv[s] = make_double2(obs[2 * (size_t)i], obs[2 * (size_t)i + 1]);
      L.cam[s] = cam_idx[i];
      L.lm[s] = l;
      if (is_long[i]) {
        const int lane = s & 63;
        L.meta[s] = lane | (lane << 8) | META_REAL | META_LONG;
      } else {
        L.meta[s] = seg_first[i] | (seg_last[i] << 8) | META_REAL;
      }
    }
  // per bin: number of doubling steps the segmented scans need = ceil(log2(longest landmark))
  for (int b = 0; b < L.n_bins; ++b) {
    int mx = 1;
    for (int l = 0; l < WAVE; ++l) {
      const int m = L.meta[(size_t)b * WAVE + l];
      if ((m & META_REAL) && !(m & META_LONG)) mx = std::max(mx, ((m >> 8) & 255) - (m & 255) + 1);
    }
    int steps = 0;
    while ((1 << steps) < mx) ++steps;
    for (int l = 0; l < WAVE; ++l) L.meta[(size_t)b * WAVE + l] |= steps << META_STEPS_SHIFT;
  }
  L.cm_slot.resize(n_obs);
  L.cm_lm.resize(n_obs);
  L.cm_uv.resize(n_obs);
  {
    // slots ascend with the observation index, so a stable counting sort by camera over the
    // observations in order yields ascending slots per camera
    std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int l = 0; l < n_lms; ++l)
      for (int i = lm_off[l]; i < lm_off[l + 1]; ++i) {
        const int64_t p = pos[cam_idx[i]]++;
        L.cm_slot[p] = L.slot_of_obs[i];
        L.cm_lm[p] = l;
        L.cm_uv[p] = make_double2(obs[2 * (size_t)i], obs[2 * (size_t)i + 1]);
      }
  }
  {
    const int n_hot = std::min(n_cams, HOT_MAX);
    for (size_t s = 0; s < n_slots; ++s)
      if ((L.meta[s] & META_REAL) && rank[L.cam[s]] <= n_hot) L.meta[s] |= rank[L.cam[s]] << META_HOT_SHIFT;
    // "cold" camera-major structure for POVAR_E0_IMPLICIT_LDSACC: only the observations whose
    // Jp^T s is NOT accumulated in LDS (camera outside the HOT_ACC_MAX hottest, or a long landmark,
    // which the lm_long driver handles through q4)
    const int n_acc = hot_acc_cap(n_cams);
    L.cc_cam_item_off.assign(n_cams + 1, 0);
    for (int c = 0; c < n_cams; ++c) {
      L.cc_cam_item_off[c] = (int)L.cc_item_off.size();
      int64_t run = 0;
      for (int64_t p = cnt[c]; p < cnt[c + 1]; ++p) {
        const int s = L.cm_slot[p];
        const bool acc = rank[c] > 0 && rank[c] <= n_acc && !(L.meta[s] & META_LONG);
        if (acc) continue;
        if (run % CM_COLD_ITEM_MAX == 0) L.cc_item_off.push_back((int)L.cc_slot.size());
        L.cc_slot.push_back(s);
        L.cc_lm.push_back(L.cm_lm[p]);
        ++run;
      }
    }
    L.cc_cam_item_off[n_cams] = (int)L.cc_item_off.size();
    L.cc_item_off.push_back((int)L.cc_slot.size());
    // inverse of cc_slot: where a cold observation's scatter scalars go (Dp::q4c)
    L.cold_pos.assign(n_slots, -1);
    for (size_t p = 0; p < L.cc_slot.size(); ++p) L.cold_pos[L.cc_slot[p]] = (int)p;
    if (!L.long_lm.empty()) {
      L.c2_pos.assign(n_slots, -1);
      L.c2_range.resize(n_cams);
      for (int c = 0; c < n_cams; ++c) {
        const int first = (int)L.c2_lm.size();
        const bool acc = rank[c] > 0 && rank[c] <= n_acc;
        if (!acc)
          for (int64_t p = cnt[c]; p < cnt[c + 1]; ++p) {
            L.c2_pos[L.cm_slot[p]] = (int)L.c2_lm.size();
            L.c2_lm.push_back(L.cm_lm[p]);
          }
        L.c2_range[c] = make_int2(first, (int)L.c2_lm.size());
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// camera-chunk layout of e0_ck: upload, kernel parameters, launch
// ------------------------------------------------------------------------------------------
// locked: called by the row-placement thread -- its HIP calls go in short pieces under g_capture_mu (povar_ctx::placer_cancel)
// POVAR_CK_MAX_CAMS lowers the camera limit of the camera-chunk kernels (tests: the fall-back to e0_lpl without a 65536-camera problem)
int ck_max_cams() {
  if (const char* e = std::getenv("POVAR_CK_MAX_CAMS")) return std::max(0, std::atoi(e));
  return 65535;
}

// Returns false only on a HIP failure (allocation, copy).  A layout this kernel family cannot run -- more than 65535 cameras
// (the lane metadata keeps a popularity rank in 16 bits), a row array of 4 GiB or more (32-bit buffer offsets) -- is not
// an error: D.ready stays false, what was uploaded is released and the term loop stays on e0_lpl / e0_lpl_h.
bool ck_upload(povar_ctx* c, povar_ctx::CkDev& D, const CkLayout& K, bool locked, size_t* bytes, bool need_uv) {
  bool ok = true;
  const bool usable = K.n_uv * sizeof(double2) < (1ull << 32) && c->n_cams <= ck_max_cams();
  D.ready = false;
  if (!usable) return true;
  auto guarded = [&](auto&& fn) {
    if (locked) {
      std::lock_guard<std::mutex> lk(g_capture_mu);
      fn();
    } else {
      fn();
    }
  };
  auto up = [&](auto& buf, const auto& v) {
    if (!ok) return;
    guarded([&] { ok = buf.alloc(std::max<size_t>(v.size(), 1), bytes) == hipSuccess; });
    const size_t piece = ((size_t)8 << 20) / sizeof(v[0]);  // 8 MB per copy: a capture waits a millisecond at most
    for (size_t at = 0; ok && at < v.size() && !(locked && c->placer_cancel.load()); at += piece)
      guarded([&] {
        ok = hipMemcpy(buf.p + at, v.data() + at, std::min(piece, v.size() - at) * sizeof(v[0]), hipMemcpyHostToDevice) == hipSuccess;
      });
  };
  std::vector<int2> meta(K.lane_cam.size());
  for (size_t i = 0; i < meta.size(); ++i) {
    const int sg = K.lane_seg[i];
    meta[i] = make_int2(K.lane_cam[i] < 0 ? -1 : (K.lane_cam[i] | ((sg & 63) << 16) | (((sg >> 8) & 63) << 22)), K.lane_acc[i]);
  }
  D.packed = need_uv && K.packed;
  if (D.packed) up(D.uvp, K.uvp);  // 8 bytes per observation where every image point is a six-decimal number (ck_pack_uv)
  else if (need_uv) up(D.uv, K.uv);  // (step 2's operator does not read the image coordinates)
  up(D.li, K.li); up(D.src, K.src); up(D.tile, K.tile); up(D.lane_meta, meta);
  up(D.bt_off, K.bt_off); up(D.slot_rec, K.slot_rec); up(D.part_range, K.part_range);
  if (c->det_ck) { up(D.lcnt, K.lcnt_log2); up(D.tick, K.tick); }
  D.cold_q = K.cold_q;
  if (K.cold_q) up(D.cpos, K.cpos);
  if (ok) guarded([&] { ok = D.part.alloc((size_t)std::max(K.n_part_rec, 1) * 12, bytes) == hipSuccess; });
  if (ok && c->opt.robust_norm && !need_uv)  // (step 2's kernel reads the weights in chunk order; step 1's recomputes them)
    guarded([&] { ok = D.w.alloc(std::max<size_t>(K.src.size(), 1), bytes) == hipSuccess; });  // (padded like the rows)
  D.nb = K.nb; D.slots = K.slots; D.n_part_rec = K.n_part_rec; D.max_acc = K.max_acc; D.max_tiles_bt = K.max_tiles_bt;
  D.stride = K.stride; D.n_capped_obs = K.n_capped_obs;
  D.rows = K.rows; D.li_rows = K.li_rows; D.n_chunks = K.n_chunks; D.n_cold_chunks = K.n_cold_chunks;
  D.w_lin_id = -1;
  D.ready = ok && !(locked && c->placer_cancel.load());
  if (!ok) D.release();
  return ok;
}

// ------------------------------------------------------------------------------------------
// resident power series (series_res): upload, kernel parameters, launch
// ------------------------------------------------------------------------------------------
int res_upload(povar_ctx* c, const ResLayout& R) {
  povar_ctx::ResDev& D = c->res;
  int rc = 0;
  if ((rc = upload(D.lane_cam, R.lane_cam, c)) || (rc = upload(D.lane_seg, R.lane_seg, c)) ||
      (rc = upload(D.uv, R.uv, c)) || (rc = upload(D.lslot, R.lslot, c)) || (rc = upload(D.oslot, R.oslot, c)) ||
      (rc = upload(D.wave_h, R.wave_h, c)) || (rc = upload(D.lm_off, R.lm_off, c)) || (rc = upload(D.lm_id, R.lm_id, c)) ||
      (rc = upload(D.cam_off, R.cam_off, c)) || (rc = upload(D.cam_id, R.cam_id, c)) || (rc = upload(D.cam_zi, R.cam_zi, c)) ||
      (rc = upload(D.own_off, R.own_off, c)) || (rc = upload(D.own_cam, R.own_cam, c)) || (rc = upload(D.own_zi, R.own_zi, c)) ||
      (rc = upload(D.own_q, R.own_q, c)) || (rc = upload(D.oq_off, R.oq_off, c)) || (rc = upload(D.oq_rec, R.oq_rec, c)))
    return rc;
  // granule buffers: tag 0 everywhere (no launch has the number 0), the launch counter starts at 1
  const size_t n_part = (size_t)std::max(R.n_rec, 1) * 12, n_z = (size_t)c->n_cams * 12, n_nrm = (size_t)RES_MAX_WG * 2 * 2;  // (two halves: the norms are double-buffered by term parity)
  if (n_part * sizeof(uint4) >= (1ull << 32)) return fail(-1, "resident series: partial records exceed a buffer descriptor");
  HIP_TRY(D.part.alloc(n_part, &c->bytes));
  HIP_TRY(D.zbuf.alloc(n_z, &c->bytes));
  HIP_TRY(D.nrm.alloc(n_nrm, &c->bytes));
  HIP_TRY(D.launch.alloc(4, &c->bytes));
  HIP_TRY(hipMemset(D.part.p, 0, n_part * sizeof(uint4)));
  HIP_TRY(hipMemset(D.zbuf.p, 0, n_z * sizeof(uint4)));
  HIP_TRY(hipMemset(D.nrm.p, 0, n_nrm * sizeof(uint4)));
  const unsigned one[4] = {1u, 0u, 0u, 0u};
  HIP_TRY(hipMemcpy(D.launch.p, one, sizeof(one), hipMemcpyHostToDevice));
  D.W = R.W; D.NW = R.NW; D.H = R.H; D.R = R.R; D.LS = R.LS; D.n_rec = R.n_rec; D.max_lm = R.max_lm; D.max_cam = R.max_cam;
  D.max_oq = R.max_oq; D.max_own = R.max_own; D.max_chunks = R.max_chunks; D.order = R.order; D.lds_bytes = R.lds_bytes;
  D.ready = true;
  return 0;
}

// the layout for a context: the lightest instantiation that holds it (fewest rows in registers first)
void res_build_for(int n_cams, int n_lms, const int32_t* lm_off, const int32_t* cam_idx, const double* obs,
                   const std::vector<int>& rank1, const std::vector<int>& slot_of_obs, int wgs, ResLayout& R) {
  build_res(n_cams, n_lms, lm_off, cam_idx, obs, rank1, slot_of_obs, wgs, 16, 1, 1, 2, 1, R);
  if (R.fits) return;
  build_res(n_cams, n_lms, lm_off, cam_idx, obs, rank1, slot_of_obs, wgs, 8, 2, 1, 4, 2, R);
}

// The placed rows (povar_ctx::placer) replace the natural order.  Only between linearisations: everything lane-ordered
// that a linearisation leaves behind (V2::lml / lsc / w, the landmark records) belongs to the row order it was built on;
// the landmark mirror V2::lmx is regathered on demand.  wait: block until the host thread is done.
// Returns 1 when the rows were swapped in.
int swap_in_placed_rows(povar_ctx* c, bool wait) {
  if (c->placement != 2) return 0;
  if (!wait && c->placer_state.load(std::memory_order_acquire) < 2) return 0;
  if (c->placer.joinable()) c->placer.join();
  if (c->placer_state.load(std::memory_order_acquire) != 2) {  // the build or an upload failed: stay on the natural order
    c->pl_uv.release(); c->pl_cw.release(); c->pl_cpos.release(); c->pl_lm_pos.release(); c->pl_lm_of.release(); c->pl_of_slot.release();
    c->pl_c3_src.release();
    c->pl_ck.release(); c->pl_ckh.release();
    c->placement = 0;
    return 0;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));  // nothing in flight reads the old rows
  std::swap(c->v2_uv, c->pl_uv); std::swap(c->v2_cw, c->pl_cw); std::swap(c->v2_cpos, c->pl_cpos);
  std::swap(c->v2_lm_pos, c->pl_lm_pos); std::swap(c->v2_lm_of, c->pl_lm_of); std::swap(c->v2_of_slot, c->pl_of_slot);
  std::swap(c->c3_src, c->pl_c3_src);  // ldsacc_dp takes it from the context at every launch
  c->pl_uv.release(); c->pl_cw.release(); c->pl_cpos.release(); c->pl_lm_pos.release(); c->pl_lm_of.release(); c->pl_of_slot.release();
  c->pl_c3_src.release();
  V2& v = c->d.v2;
  v.uv = c->v2_uv.p; v.cw = c->v2_cw.p; v.cpos = c->v2_cpos.p; v.lm_pos = c->v2_lm_pos.p; v.lm_of = c->v2_lm_of.p;
  v.of_slot = c->v2_of_slot.p;
  c->lmx_ver = 0;                          // lane-ordered landmark mirror: regather
  c->lml_lin_id = c->lsc_lin_id = -1;
  // the camera-chunk layout belongs to the row order it was derived from
  c->ck.release();
  if (c->pl_ck.ready) std::swap(c->ck, c->pl_ck);
  c->pl_ck.release();
  c->ckh.release();
  if (c->pl_ckh.ready) std::swap(c->ckh, c->pl_ckh);
  c->pl_ckh.release();
  c->ck_tuned = c->ckh_tuned = false;  // (the choice between the E0 kernels is timed again on the new rows)
  c->placement = 3;
  return 1;
}

int check_ctx(povar_ctx* c) {
  if (!c) return fail(-1, "null context");
  HIP_TRY(hipSetDevice(c->opt.device));
  return 0;
}

// Small results come back through one pinned block: an asynchronous copy into pageable memory is staged by the
// runtime (a synchronisation per copy), two of those per API call were most of its latency.
// Layout of the block: [0, 16) the four flags, [64, 64 + 8 * 16) scalars, [256, ...) one 12 n_cams vector.
int ensure_pin(povar_ctx* c) {
  if (c->pin) return 0;
  c->pin_bytes = 256 + sizeof(double) * 12 * (size_t)std::max(c->n_cams, 1);
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->pin), c->pin_bytes, hipHostMallocDefault));
  return 0;
}

int read_scal_flags(povar_ctx* c, double* h, int n, int (&f)[4]) {
  if (int rc = ensure_pin(c)) return rc;
  if (n > 0) HIP_TRY(hipMemcpyAsync(c->pin + 64, c->scal.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(c->pin, c->flags.p, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::memcpy(f, c->pin, sizeof(int) * 4);
  if (n > 0) std::memcpy(h, c->pin + 64, sizeof(double) * n);
  return 0;
}

int read_flags(povar_ctx* c, int (&f)[4]) { return read_scal_flags(c, nullptr, 0, f); }

int read_scal(povar_ctx* c, double* h, int n) {
  if (int rc = ensure_pin(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->pin + 64, c->scal.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::memcpy(h, c->pin + 64, sizeof(double) * n);
  return 0;
}

int write_cam_vector(povar_ctx* c, double* dst, const double* in, size_t n) {  // completes with the caller's next sync
  if (int rc = ensure_pin(c)) return rc;
  std::memcpy(c->pin + 256, in, sizeof(double) * n);
  HIP_TRY(hipMemcpyAsync(dst, c->pin + 256, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  return 0;
}

int read_cam_vector(povar_ctx* c, double* out, const double* src, size_t n) {
  if (int rc = ensure_pin(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->pin + 256, src, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::memcpy(out, c->pin + 256, sizeof(double) * n);
  return 0;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

const char* povar_last_error(void) { return g_err.c_str(); }

int povar_device_count(void) {
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  return n;
}

int povar_device_cu_count(int32_t device) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  return prop.multiProcessorCount;
}

int povar_shard_range(int32_t n_lms, const int32_t* lm_offsets, int32_t world, int32_t rank,
                      int32_t* lm_begin, int32_t* lm_end) {
  if (!lm_offsets || world < 1 || rank < 0 || rank >= world) return fail(-1, "bad shard arguments");
  // contiguous landmark ranges balanced by observation count (prefix sum over k_l): boundary r is
  // the first landmark whose observations start at or after r/world of the total.
  const int64_t n_obs = lm_offsets[n_lms];
  auto bound = [&](int r) -> int32_t {
    if (r <= 0) return 0;
    if (r >= world) return n_lms;
    const int64_t target = n_obs * (int64_t)r / world;
    return (int32_t)(std::lower_bound(lm_offsets, lm_offsets + n_lms + 1, (int32_t)target) - lm_offsets);
  };
  *lm_begin = bound(rank);
  *lm_end = bound(rank + 1);
  return 0;
}

// inside povar_create, once the context exists: a failing HIP call releases everything allocated so far
#define HIP_TRY_C(expr)                                                                     \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      povar_destroy(c);                                                                     \
      return fail(-(int)e_ - 1000, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    }                                                                                       \
  } while (0)
int povar_create(povar_ctx** out, int32_t n_cams, int32_t n_lms, int64_t n_obs,
                 const int32_t* lm_offsets, const int32_t* cam_idx, const double* obs,
                 const povar_options* options) {
  if (!out || !lm_offsets || !cam_idx || !obs || !options) return fail(-1, "null argument");
  if (n_cams <= 0 || n_lms <= 0 || n_obs <= 0 || lm_offsets[0] != 0 || lm_offsets[n_lms] != n_obs)
    return fail(-1, "invalid problem sizes");
  const bool timing = std::getenv("POVAR_LAYOUT_TIMING") != nullptr;
  const auto t_create = std::chrono::steady_clock::now();
  auto t_last = t_create;
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[povar_create] %-26s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  for (int l = 0; l < n_lms; ++l) {
    if (lm_offsets[l + 1] < lm_offsets[l]) return fail(-1, "lm_offsets not monotone");
    for (int i = lm_offsets[l]; i < lm_offsets[l + 1]; ++i) {
      if (cam_idx[i] < 0 || cam_idx[i] >= n_cams) return fail(-1, "camera index out of range");
      // duplicate (camera, landmark) pairs abort the reference loader (bal_problem.cpp:227)
      if (i > lm_offsets[l] && cam_idx[i] <= cam_idx[i - 1])
        return fail(-1, "camera indices of a landmark must be strictly ascending");
    }
  }
  lap("argument checks");
  int n_dev = 0;
  HIP_TRY(hipGetDeviceCount(&n_dev));
  if (n_dev <= 0) return fail(-2, "no HIP device: the MI355X path has no CPU fallback");
  HIP_TRY(hipSetDevice(options->device));

  povar_ctx* c = new povar_ctx();
  c->opt = *options;
  c->n_cams = n_cams;
  c->n_lms = n_lms;
  c->n_obs = n_obs;
  c->lm_off.assign(lm_offsets, lm_offsets + n_lms + 1);
  for (int l = 0; l < n_lms; ++l) c->has_empty_lm |= lm_offsets[l + 1] == lm_offsets[l];
  // POVAR_CU_MASK=<first>-<last>: the context's stream runs on that range of CUs only.  For several contexts that share ONE
  // device and wait for each other inside kernels (the peer-to-peer term exchange with two ranks on a one-GPU box): an
  // E0 workgroup takes a CU's whole register file, so the bounded spin of one rank's reduce kernel on every CU kept the
  // other rank's E0 kernel off the device until the spin timed out.  Disjoint CU ranges make the ranks two half devices.
  if (const char* g = std::getenv("POVAR_CU_MASK")) {
    int first = 0, last = -1;
    hipDeviceProp_t prop;
    HIP_TRY_C(hipGetDeviceProperties(&prop, options->device));
    if (std::sscanf(g, "%d-%d", &first, &last) != 2 || first < 0 || last < first || last >= prop.multiProcessorCount) {
      povar_destroy(c);
      return fail(-1, "POVAR_CU_MASK: expected <first>-<last> inside the device's CU range");
    }
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u);
    for (int i = first; i <= last; ++i) mask[i / 32] |= 1u << (i % 32);
    HIP_TRY_C(hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()));
    c->cu_limit = last - first + 1;  // "one workgroup per CU" then means per CU of the range
  } else {
    HIP_TRY_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  }
  // POVAR_DETERMINISTIC=1: run-to-run BIT-reproducible results for a given device count (SURVEY 8(e) "fixed reduction order
  // inside a GPU"), whatever E0 mode the caller asked for, and no run-time timing decides a kernel:
  //   * linearisation, preparation and cost of both steps run in the gather mode (POVAR_E0_IMPLICIT: per-landmark wavefront
  //     scans, per-camera sums through the camera-major index -- no atomics anywhere);
  //   * the terms of the power series -- the hot path -- run e0_ck_det (step 1) / e0_ck_h_det (step 2)
  //     (povar_kernels_ck_det.hpp: the camera-chunk kernels with the landmark sums in 64-bit fixed point -- integer adds are
  //     associative -- and the accumulator adds in ticket order) + cam_cold_sum_binv[_h] (a fixed-order sum), on the landmark
  //     records and camera image the gather mode's kernels leave in lane order anyway.  POVAR_DET_CK=0: the gather form there
  //     too (3.7 x slower than the default mode on venice-1778, profiles/r05_experiments.txt; also what runs when a chunk
  //     layout does not fit);
  //   * the rows are never placed on a host thread (when they arrive would decide the bits of every later solve).
  // The default mode accumulates in LDS in arrival order and is reproducible to rounding (1e-15), like the reference's
  // mutex order.
  // The switches come from povar_options.flags (include/povar_hip.h: POVAR_FLAG_*); an environment variable that is set
  // overrides its flag (diagnosis, the forced-mode suites).
  const uint32_t fl = options->flags;
  bool want_det = (fl & POVAR_FLAG_DETERMINISTIC) != 0, det_gather = (fl & POVAR_FLAG_DET_GATHER_TERMS) != 0;
  if (const char* g = std::getenv("POVAR_DETERMINISTIC")) want_det = g[0] == '1';
  if (const char* k = std::getenv("POVAR_DET_CK")) det_gather = k[0] == '0';
  if (want_det) {
    c->opt.e0_mode = POVAR_E0_IMPLICIT;
    c->ck_auto = false;
    c->ck_variant = c->ckh_variant = 0;
    c->res_mode = 0;
    c->deterministic = true;
    c->det_ck = !det_gather;
  }
  if (fl & POVAR_FLAG_NO_GRAPH) c->use_graph = false;
  if (const char* g = std::getenv("POVAR_NO_GRAPH")) c->use_graph = !(g[0] == '1');
  if (const char* g = std::getenv("POVAR_NO_ERR_MEMO")) c->no_err_memo = g[0] == '1';
  if (const char* g = std::getenv("POVAR_GRAPH_COMM")) c->graph_with_comm = g[0] == '1';
  if (const char* g = std::getenv("POVAR_NO_FUSE")) c->fuse_binv = !(g[0] == '1');
  // A problem that gives the 256 x 16 wavefronts of the lane-per-landmark kernels less than a row each is bound by the
  // launch of the 1024-thread workgroups: the lane-per-observation kernels of round 1 are faster there (ladybug-49,
  // 31 843 observations: 127 k against 106 k terms/s; trafalgar-257, 225 911: 64.9 k against 67.0 k).
  c->use_lpl = n_obs >= 65536;
  if (const char* g = std::getenv("POVAR_E0_V1")) { c->use_lpl = !(g[0] == '1'); c->lpl_forced = true; }
  if (const char* g = std::getenv("POVAR_K1_NORMAL_EQ")) c->k1_qr = !(g[0] == '1');
  if (const char* g = std::getenv("POVAR_PREPARE_V1")) c->use_lpl_prepare = !(g[0] == '1');

  lap("device, stream");
  Layout L;
  LayoutTmp LT;
  build_layout_a(n_cams, n_lms, lm_offsets, cam_idx, L, LT);
  lap("slots, camera ranks");
  c->n_bins = L.n_bins;
  c->n_slots = L.n_bins * WAVE;
  c->n_items = (int)L.item_cam.size();
  c->n_long = (int)L.long_lm.size();
  c->n_reg_blocks = grid_for(c->n_slots, LM_BLOCK);
  c->n_cam_blocks = grid_for(n_cams, K9_CAMS);
  c->slot_of_obs = L.slot_of_obs;
  c->n_hot_acc = hot_acc_cap(n_cams);
  {
    // one 1024-thread workgroup per CU for the LDS-cached E0 kernel
    hipDeviceProp_t prop;
    HIP_TRY_C(hipGetDeviceProperties(&prop, options->device));
    int cus = std::max(c->cu_limit > 0 ? c->cu_limit : prop.multiProcessorCount, 1);
    if (const char* e = std::getenv("POVAR_E0_WGS")) cus = std::max(std::atoi(e), 1);  // tuning knob: E0 workgroups
    c->e0c_bins_per_wg = std::max((c->n_bins + cus - 1) / cus, 1);
    c->e0c_grid = (c->n_bins + c->e0c_bins_per_wg - 1) / c->e0c_bins_per_wg;
    c->n_hot = std::min(n_cams, HOT_MAX);
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_lm_cached<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HOT_MAX * HOT_REC * (int)sizeof(double2)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_tiles_cached, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HOT_ACC_MAX * (HOT_REC_T * (int)sizeof(double2) + 96)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_lm_cached_h, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HOT_ACC_MAX * (HOT_REC_H * (int)sizeof(double2) + 96)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_lm_cached<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HOT_ACC_MAX * (HOT_REC * (int)sizeof(double2) + 96)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_lpl<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lpl_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_lpl<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lpl_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_lpl_h<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lpl_lds_bytes_h(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_lpl_h<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lpl_lds_bytes_h(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)prepare_lpl<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)prep_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)prepare_lpl<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)prep_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)lpl_pass_h<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pass_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)lpl_pass_h<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pass_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)backsub_lpl_h<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)back_lds_bytes_h(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)backsub_lpl_h<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)back_lds_bytes_h(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)lpl_pass<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pass_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)lpl_pass<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)pass_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)backsub_lpl<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)back_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)backsub_lpl<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)back_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)prepare_lpl_h<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)prep_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(hipFuncSetAttribute((const void*)prepare_lpl_h<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)prep_lds_bytes(HOT_ACC_MAX)));
    HIP_TRY_C(ck_set_lds_all());
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_ck_h<16, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_ck_h<16, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_ck_h<16, 2, false, CKH_STRIDE_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES));
    HIP_TRY_C(hipFuncSetAttribute((const void*)e0_ck_h<16, 2, true, CKH_STRIDE_WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES));
  }
  // per-term E0 kernel of step 1: e0_lpl (0) or an e0_ck instantiation (POVAR_E0_CK=<variant>, povar_set_e0_kernel)
  {
    int e0k = (int)((fl & POVAR_FLAG_E0_KERNEL_MASK) >> POVAR_FLAG_E0_KERNEL_SHIFT) - 1;  // -1: timed
    if (const char* g = std::getenv("POVAR_E0_CK")) e0k = std::atoi(g);
    if (e0k >= 0 && !c->deterministic) {
      c->ck_variant = std::max(0, std::min(CK_VARIANTS, e0k));
      c->ckh_variant = c->ck_variant > 0 ? 1 : 0;  // (step 2 has one camera-chunk instantiation)
      c->ck_auto = false;
    }
  }
  const bool want_ck = c->use_lpl && std::getenv("POVAR_NO_CK") == nullptr;
  // the camera-chunk layout is cut for the instantiation that will run it (its tiles are scheduled over its wavefronts)
  const CkVariant ckv = ck_variant_info(c->ck_variant > 0 ? c->ck_variant : 1);
  int ck_nw = ckv.nw / ckv.ng, ck_hmax = CK_HMAX;  // (wavefronts of one group)
  const int ck_ng = ckv.ng;
  const bool ck_place = std::getenv("POVAR_CK_NOPLACE") == nullptr;  // LDS bank placement of the chunk rows (ck_layout.hpp)
  const bool ck_pack = !(fl & POVAR_FLAG_NO_PACKED_ROWS);            // packed image points where they pack (POVAR_CK_PACK=0 overrides inside build_ck)
  if (const char* e = std::getenv("POVAR_CK_HMAX")) ck_hmax = std::min(CK_HMAX, std::max(1, std::atoi(e)));
  // (step 1's layout: batches cut for the kernel that runs them; e0_ck leaves q of its cold observations in the parent's cold view)
  // (not with the HUBER norm: e0_ck<..., ROBUST = true> has no cold loop -- povar_kernels_ck.hpp)
  const CkShape ck_shape1 = c->det_ck ? ck_shape_det()
                            : (std::getenv("POVAR_CK_COLD_RECORDS") || options->robust_norm == POVAR_NORM_HUBER) ? CkShape()
                            : ck_shape_step1();
  const CkShape ck_shape2 = c->det_ck ? ck_shape_step2_det() : ck_shape_step2();

  // The arrays of the lane-per-observation kernels (part B) are built on a second host thread while this one builds
  // and uploads the lane-per-landmark layout: the two only share the slot numbers and camera ranks of part A.
  std::atomic<bool> part_b_failed{false};
  std::thread part_b([&]() {
    try {
      build_layout_b(n_cams, n_lms, lm_offsets, cam_idx, obs, L, LT);
    } catch (...) {  // (an allocation failure of this thread must not terminate the caller's process)
      part_b_failed.store(true);
    }
  });
  struct Joiner {  // every early return below must wait for the thread
    std::thread& t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } joiner{part_b};
  size_t n_cold_lpl = 0, n_cold_q = 0;
  {
    // lane-per-landmark layout of e0_lpl (lpl_layout.hpp)
    LplLayout V;
    int place_mode = n_obs >= (1 << 20) ? 2 : 1;  // 0 none, 1 in this call, 2 on a host thread
    bool place_forced = false;
    if (const uint32_t pf = (fl & POVAR_FLAG_PLACEMENT_MASK) >> POVAR_FLAG_PLACEMENT_SHIFT) { place_mode = pf == 3 ? 0 : (int)pf; place_forced = true; }
    if (std::getenv("POVAR_LPL_NOPLACE")) place_mode = 0;
    if (const char* e = std::getenv("POVAR_LPL_PLACE")) { place_mode = e[0] == 'n' ? 0 : e[0] == 's' ? 1 : e[0] == 'a' ? 2 : place_mode; place_forced = true; }
    // POVAR_DETERMINISTIC: never on a host thread -- WHEN the placed rows (and the chunk layout cut from them: another, equally
    // fixed summation order) arrive would depend on the host's timing, and with it the bits of every later solve.  None at
    // all unless asked for: the gather-mode kernels do not read these rows, e0_ck_det does not care about their order.
    if (c->deterministic) place_mode = place_mode == 1 && place_forced ? 1 : 0;
    build_lpl(n_cams, n_lms, lm_offsets, cam_idx, obs, L.cam_hot, L.slot_of_obs, (size_t)c->n_slots, c->e0c_grid,
              c->n_hot_acc, V, place_mode == 1);
    lap("build_lpl (lane/landmark)");
    c->placement = place_mode;
    // The rows are placed on a host thread, and the chunk layouts of the placed rows arrive with them -- half a second later
    // on venice-1778, i.e. after the first two hundred LM iterations.  e0_ck does not care which order the
    // lane-per-landmark rows are in (its own bank placement is what counts: 16.0 k terms/s on the natural rows, 15.9 k
    // without its placement on either; profiles/r05_experiments.txt), so step 1's layout is built HERE from the natural
    // rows -- before the host thread starts: the two would share the CPUs -- for e0_ck from the first solve on (0.09 s of
    // povar_create; `bal` on venice: 96 -> 62 us per term).  POVAR_CKH_EARLY=1: step 2's instance likewise.
    std::unique_ptr<CkLayout> ck_nat, ckh_nat;
    if (place_mode == 2 && want_ck && !V.tile.empty()) {
      const auto tk = std::chrono::steady_clock::now();
      ck_nat.reset(new CkLayout());
      build_ck(V, n_cams, c->e0c_grid, L.hot_cams, ck_nw, *ck_nat, ck_place, ck_hmax, ck_ng, ck_shape1, ck_pack);
      c->ck.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk).count();
      if (std::getenv("POVAR_CKH_EARLY") != nullptr) {
        ckh_nat.reset(new CkLayout());
        build_ck(V, n_cams, c->e0c_grid, L.hot_cams, 16, *ckh_nat, ck_place, ck_hmax, 1, ck_shape2);
      }
      lap("camera-chunk layout(s) from the natural rows");
    }
    if (place_mode == 2 && !V.tile.empty()) {
      // the same builder again, with the placement, on copies of the caller's arrays (they need not outlive this call)
      struct Job {
        std::vector<int32_t> lm_off, cam_idx;
        std::vector<double> obs;
        std::vector<int> rank1, slot_of_obs, cam_of_rank;
      };
      auto job = std::make_shared<Job>();
      job->lm_off.assign(lm_offsets, lm_offsets + n_lms + 1);
      job->cam_idx.assign(cam_idx, cam_idx + n_obs);
      job->obs.assign(obs, obs + 2 * n_obs);
      job->rank1 = L.cam_hot;
      job->slot_of_obs = L.slot_of_obs;
      job->cam_of_rank = L.hot_cams;
      const int64_t rows = V.rows;
      const size_t n_tiles = V.tile.size();
      const int dev = options->device, grid = c->e0c_grid, n_acc = c->n_hot_acc;
      const size_t n_slots = (size_t)c->n_slots;
      c->placer_state.store(1);
      c->placer = std::thread([c, job, rows, n_tiles, dev, grid, n_acc, n_slots, n_cams, n_lms, want_ck, ck_nw, ck_hmax, ck_ng, ck_place, ck_pack, ck_shape1, ck_shape2]() {
        const auto t0 = std::chrono::steady_clock::now();
        LplLayout P;
        bool built = true;
        try {  // an allocation failure of this thread must not terminate the caller's process: stay on the natural order
          build_lpl(n_cams, n_lms, job->lm_off.data(), job->cam_idx.data(), job->obs.data(), job->rank1, job->slot_of_obs,
                    n_slots, grid, n_acc, P, true, &c->placer_cancel);
        } catch (...) {
          built = false;
        }
        bool ok = built && !c->placer_cancel.load() && P.rows == rows && P.tile.size() == n_tiles;
        if (ok) {
          std::lock_guard<std::mutex> lk(g_capture_mu);
          ok = hipSetDevice(dev) == hipSuccess;
        }
        auto up = [&](auto& buf, const auto& v) {
          if (!ok) return;
          {
            std::lock_guard<std::mutex> lk(g_capture_mu);
            ok = buf.alloc(std::max<size_t>(v.size(), 1), &c->pl_bytes) == hipSuccess;
          }
          const size_t piece = ((size_t)8 << 20) / sizeof(v[0]);  // 8 MB per copy: a capture waits a millisecond at most
          for (size_t at = 0; ok && at < v.size() && !c->placer_cancel.load(); at += piece) {
            std::lock_guard<std::mutex> lk(g_capture_mu);
            ok = hipMemcpy(buf.p + at, v.data() + at, std::min(piece, v.size() - at) * sizeof(v[0]), hipMemcpyHostToDevice) == hipSuccess;
          }
        };
        up(c->pl_uv, P.uv); up(c->pl_cw, P.cw); up(c->pl_cpos, P.cpos);
        up(c->pl_lm_pos, P.lm_pos); up(c->pl_lm_of, P.lm_of); up(c->pl_of_slot, P.of_slot);
        up(c->pl_c3_src, P.cold_src);
        if (ok && want_ck) {  // the camera-chunk layout of the placed rows (a failure here only leaves e0_lpl in charge)
          try {
            const auto tk = std::chrono::steady_clock::now();
            CkLayout K;
            build_ck(P, n_cams, grid, job->cam_of_rank, ck_nw, K, ck_place, ck_hmax, ck_ng, ck_shape1, ck_pack);
            c->pl_ck.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk).count();
            if (!c->placer_cancel.load()) ck_upload(c, c->pl_ck, K, true, &c->pl_bytes);
            if (!c->placer_cancel.load()) {  // step 2's instance
              CkLayout KH;
              build_ck(P, n_cams, grid, job->cam_of_rank, 16, KH, ck_place, ck_hmax, 1, ck_shape2);
              if (!c->placer_cancel.load()) ck_upload(c, c->pl_ckh, KH, true, &c->pl_bytes, false);
            }
          } catch (...) {
            c->pl_ck.ready = false;
            c->pl_ckh.ready = false;
          }
        }
        c->placement_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        c->placer_state.store(ok ? 2 : 3, std::memory_order_release);
      });
      lap("row placement handed to a host thread");
    }
    if (V.max_slots > c->n_hot_acc) { povar_destroy(c); return fail(-1, "lpl layout: workgroup camera set exceeds the LDS capacity"); }
    c->v2_rows = V.rows;
    c->v2_max_slots = V.max_slots;
    c->v2_n_global = V.n_global;
    c->v2_strategy = V.strategy;
    c->v2_n_tail = V.n_tail;
    c->n_cold3 = (int64_t)V.cold_lm.size();
    if (int rc = upload(c->v2_uv, V.uv, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_cw, V.cw, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_cpos, V.cpos, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_lm_pos, V.lm_pos, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_lm_of, V.lm_of, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_of_slot, V.of_slot, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_tile, V.tile, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_seg, V.seg, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_wg_tile_off, V.wg_tile_off, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_wg_cam_off, V.wg_cam_off, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_wg_cams, V.wg_cams, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_wg_slot_rec, V.wg_slot_rec, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->v2_part_range, V.part_range, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->c3_lm, V.cold_lm, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->c3_range, V.cold_range, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->c3_src, V.cold_src, c)) { povar_destroy(c); return rc; }
    n_cold_q = (size_t)V.cold_rows * WAVE;
    c->q_rows = (double)V.cold_lm.size() >= 0.20 * (double)std::max<int64_t>(n_obs, 1);
    if (const char* e = std::getenv("POVAR_COLD_Q_ROWS")) c->q_rows = e[0] == '1';
    const int nt = (int)V.tile.size();
    HIP_TRY_C(c->v2_lmrec.alloc((size_t)std::max(nt, 1) * LPL_REC_H * WAVE, &c->bytes));  // 9 entries used by step 1
    HIP_TRY_C(c->v2_lmx.alloc((size_t)std::max(nt, 1) * WAVE, &c->bytes));
    HIP_TRY_C(c->v2_lml.alloc((size_t)std::max(nt, 1) * WAVE, &c->bytes));
    HIP_TRY_C(c->v2_lsc.alloc((size_t)std::max(nt, 1) * WAVE, &c->bytes));
    HIP_TRY_C(c->v2_part.alloc((size_t)std::max(V.n_part_rec, 1) * 12, &c->bytes));
    HIP_TRY_C(c->c3_h.alloc(4 * std::max<size_t>(V.cold_lm.size(), 1), &c->bytes));
    if (options->robust_norm) HIP_TRY_C(c->v2_w.alloc((size_t)std::max<int64_t>(c->v2_rows, 1) * WAVE, &c->bytes));
    n_cold_lpl = V.cold_lm.size();
    if (want_ck && !V.tile.empty()) {
      if (int rc = upload(c->ck_zero_range, std::vector<int2>((size_t)n_cams, make_int2(0, 0)), c)) { povar_destroy(c); return rc; }
      if (place_mode == 2) {
        if (!ck_upload(c, c->ck, *ck_nat, false, &c->bytes)) { povar_destroy(c); return fail(-1, "camera-chunk layout: upload failed"); }
        if (ckh_nat && !ck_upload(c, c->ckh, *ckh_nat, false, &c->bytes, false)) { povar_destroy(c); return fail(-1, "camera-chunk layout (step 2): upload failed"); }
        lap("camera-chunk layouts (natural rows): uploads");
      } else {  // (step 2's, and both of the placed rows, come from the host thread with place_mode 2)
        const auto tk = std::chrono::steady_clock::now();
        CkLayout K;
        build_ck(V, n_cams, c->e0c_grid, L.hot_cams, ck_nw, K, ck_place, ck_hmax, ck_ng, ck_shape1, ck_pack);
        c->ck.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk).count();
        if (!ck_upload(c, c->ck, K, false, &c->bytes)) { povar_destroy(c); return fail(-1, "camera-chunk layout: upload failed"); }
        CkLayout KH;  // step 2's instance: 64 bytes of LDS per landmark slot, no image coordinates
        build_ck(V, n_cams, c->e0c_grid, L.hot_cams, 16, KH, ck_place, ck_hmax, 1, ck_shape2);
        if (!ck_upload(c, c->ckh, KH, false, &c->bytes, false)) { povar_destroy(c); return fail(-1, "camera-chunk layout (step 2): upload failed"); }
        lap("camera-chunk layouts");
      }
    }
    c->d.v2 = V2{c->v2_uv.p, c->v2_cw.p, c->v2_cpos.p, c->v2_w.p, c->v2_tile.p, c->v2_seg.p, c->v2_lmrec.p,
                 c->v2_lm_of.p, c->v2_lmx.p, c->v2_lml.p, c->v2_lsc.p, c->v2_lm_pos.p, c->v2_of_slot.p, c->v2_wg_tile_off.p, c->v2_wg_cam_off.p, c->v2_wg_cams.p,
                 c->v2_wg_slot_rec.p, nt, V.hubs};
  }
  lap("uploads (lane/landmark)");
  {
    // resident power series (res_layout.hpp): for contexts whose observations fit the lanes' registers.  POVAR_RES=0|1
    // forces the choice (default: timed against the per-term kernels at the first series), POVAR_RES_WGS the workgroups,
    // POVAR_RES_OBS_PER_WG the observations a workgroup gets on small problems before all CUs are used.
    if (const uint32_t sf = (fl & POVAR_FLAG_SERIES_KERNEL_MASK) >> POVAR_FLAG_SERIES_KERNEL_SHIFT; sf && !c->deterministic) c->res_mode = (int)sf - 1;
    if (const char* e = std::getenv("POVAR_RES"); e && !c->deterministic) c->res_mode = e[0] == '1' ? 1 : 0;
    if (const char* e = std::getenv("POVAR_RES_SPIN")) c->res_spin_limit = (unsigned)std::max(1, std::atoi(e));
    // (measured, profiles/r05_res_term_times.txt: ahead of the per-term kernels up to a shard of 313 k observations, behind them
    // on one of 625 k, where the partial records -- 21.7 MB written and read per term -- are the term)
    int64_t max_obs = 400000;
    if (const char* e = std::getenv("POVAR_RES_MAX_OBS")) max_obs = std::atoll(e);
    if (c->res_mode != 0 && n_obs <= max_obs) {
      const auto tr = std::chrono::steady_clock::now();
      hipDeviceProp_t prop;
      HIP_TRY_C(hipGetDeviceProperties(&prop, options->device));
      const int cus = std::min(std::max(c->cu_limit > 0 ? c->cu_limit : prop.multiProcessorCount, 1), RES_MAX_WG);
      int per_wg = 256;  // (ladybug-49: 32 workgroups 8.7, 63: 7.5, 125: 7.3 us per term -- the phases of a term are the workgroup's size)
      if (const char* e = std::getenv("POVAR_RES_OBS_PER_WG")) per_wg = std::max(64, std::atoi(e));
      int wgs = (int)std::min<int64_t>(cus, std::max<int64_t>(8, (n_obs + per_wg - 1) / per_wg));
      if (const char* e = std::getenv("POVAR_RES_WGS")) wgs = std::max(1, std::min(std::atoi(e), cus));
      HIP_TRY_C(res_set_lds_all());
      ResLayout R;
      res_build_for(n_cams, n_lms, lm_offsets, cam_idx, obs, L.cam_hot, L.slot_of_obs, wgs, R);
      if (R.fits && res_variant_exists(R.NW, R.H, R.R, R.LS)) {
        if (int rc = res_upload(c, R)) { povar_destroy(c); return rc; }
        c->res.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr).count();
      }
      lap("resident-series layout");
    }
  }
  part_b.join();
  if (part_b_failed.load()) { povar_destroy(c); return fail(-4, "out of host memory while building the lane-per-observation layout"); }
  lap("wait for the lane/obs arrays");
  int rc = 0;
  if ((rc = upload(c->uv, L.uv, c)) || (rc = upload(c->cam, L.cam, c)) || (rc = upload(c->lm, L.lm, c)) ||
      (rc = upload(c->meta, L.meta, c)) || (rc = upload(c->hot_cams, L.hot_cams, c)) || (rc = upload(c->cam_hot, L.cam_hot, c)) ||
      (rc = upload(c->cc_slot, L.cc_slot, c)) || (rc = upload(c->cc_lm, L.cc_lm, c)) ||
      (rc = upload(c->cc_item_off, L.cc_item_off, c)) || (rc = upload(c->cc_cam_item_off, L.cc_cam_item_off, c)) || (rc = upload(c->long_lm, L.long_lm, c)) ||
      (rc = upload(c->long_first, L.long_first, c)) || (rc = upload(c->long_cnt, L.long_cnt, c)) ||
      (rc = upload(c->cm_slot, L.cm_slot, c)) || (rc = upload(c->cm_lm, L.cm_lm, c)) ||
      (rc = upload(c->cm_uv, L.cm_uv, c)) || (rc = upload(c->item_off, L.item_off, c)) ||
      (rc = upload(c->item_cam, L.item_cam, c)) || (rc = upload(c->cam_item_off, L.cam_item_off, c))) {
    povar_destroy(c);
    return rc;
  }
  lap("uploads (lane/obs)");
  const size_t nc = n_cams, nl = n_lms, ns = c->n_slots, ni = std::max(c->n_items, 1);
  const size_t n_part = (size_t)(c->n_reg_blocks + c->n_long) * 4;
#define ALLOC(buf, count)                                  \
  do {                                                     \
    hipError_t e_ = c->buf.alloc((count), &c->bytes);      \
    if (e_ != hipSuccess) {                                \
      povar_destroy(c);                                    \
      return fail(-(int)e_ - 1000, "hipMalloc " #buf);     \
    }                                                      \
  } while (0)
  ALLOC(cams4, 3 * nc); ALLOC(cams_lin4, 3 * nc); ALLOC(cams_bak4, 3 * nc);
  ALLOC(lms4, nl); ALLOC(lms_lin4, nl); ALLOC(lms_bak4, nl); ALLOC(jl_scale4, nl);
  ALLOC(hll_inv, 9 * nl); ALLOC(lmrec, 16 * nl);
  ALLOC(sw, ns); ALLOC(rres, ns); ALLOC(q4, ns);
  ALLOC(sigma, 12 * nc); ALLOC(diag2, 12 * nc); ALLOC(G, 40 * nc); ALLOC(binv, 144 * nc);
  ALLOC(b, 12 * nc); ALLOC(tmp, 12 * nc); ALLOC(accum, 12 * nc); ALLOC(z, 12 * nc); ALLOC(y, 12 * nc);
  ALLOC(inc, 12 * nc);
  ALLOC(item_part, 12 * ni); ALLOC(item_partG, 40 * ni); ALLOC(cm_h, 4 * (size_t)n_obs); ALLOC(ncw, 13 * nc);
  c->n_cold = (int64_t)L.cc_slot.size();
  c->n_cold_items = (int)L.cc_item_off.size() - 1;
  {
    std::vector<int2> range(n_cams);
    for (int k = 0; k < n_cams; ++k)
      range[k] = make_int2(L.cc_item_off[L.cc_cam_item_off[k]], L.cc_item_off[L.cc_cam_item_off[k + 1]]);
    if (int rc = upload(c->cc_cam_range, range, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->cold_pos, L.cold_pos, c)) { povar_destroy(c); return rc; }
    if (!L.long_lm.empty()) {
      c->n_cold2 = (int64_t)L.c2_lm.size();
      if (int rc = upload(c->c2_lm, L.c2_lm, c)) { povar_destroy(c); return rc; }
      if (int rc = upload(c->c2_pos, L.c2_pos, c)) { povar_destroy(c); return rc; }
      if (int rc = upload(c->c2_range, L.c2_range, c)) { povar_destroy(c); return rc; }
      HIP_TRY_C(c->c2_h.alloc(4 * std::max<size_t>(L.c2_lm.size(), 1), &c->bytes));
      // knob POVAR_LONG_SEPARATE: keep the lm_long kernel (old lane-per-observation kernels only; e0_lpl has no
      // long/short distinction and always uses this cold view)
      c->long_in_kernel = c->use_lpl || std::getenv("POVAR_LONG_SEPARATE") == nullptr;
    }
  }
  {
    std::vector<int> s0(n_lms, 0), cnt(n_lms, 0);
    for (int l = 0; l < n_lms; ++l) {
      cnt[l] = lm_offsets[l + 1] - lm_offsets[l];
      s0[l] = cnt[l] > 0 ? L.slot_of_obs[lm_offsets[l]] : 0;
    }
    if (int rc = upload(c->lm_slot0, s0, c)) { povar_destroy(c); return rc; }
    if (int rc = upload(c->lm_cnt_dev, cnt, c)) { povar_destroy(c); return rc; }
    c->d.lm_slot0 = c->lm_slot0.p;
    c->d.lm_cnt = c->lm_cnt_dev.p;
  }
  lap("uploads, allocations (lane/obs)");
  // scatter scalars of the cold observations: one buffer, sized for the largest of the cold views
  HIP_TRY_C(c->q4c.alloc(std::max<size_t>(std::max(std::max(std::max(L.cc_slot.size(), L.c2_lm.size()), n_cold_lpl), n_cold_q), 1), &c->bytes));
  ALLOC(cc_h, 4 * std::max<size_t>(L.cc_slot.size(), 1)); ALLOC(cc_part, 12 * (size_t)std::max(c->n_cold_items, 1));
  ALLOC(hot_part, (size_t)c->e0c_grid * c->n_hot_acc * 12);
  ALLOC(hot_rec, (size_t)std::max(n_cams, HOT_MAX) * HOT_REC_STRIDE);  // every camera, in popularity order
  ALLOC(zimg, (size_t)n_cams * 12);  // z alone, by rank: what e0_ck gathers Z from (Dp::zimg)
  ALLOC(norm_part, 2 * (size_t)std::max(c->n_cam_blocks, n_cams)); ALLOC(norms, 4); ALLOC(flags, 4);
  ALLOC(part, n_part * 2 + 8 * 1024); ALLOC(scal, 8);  // + one slot set per workgroup of the lane-per-landmark kernels
  ALLOC(stage, std::max(3 * nl, 144 * nc));
#undef ALLOC
  HIP_TRY_C(hipMemsetAsync(c->flags.p, 0, sizeof(int) * 4, c->stream));
  HIP_TRY_C(hipMemsetAsync(c->lms4.p, 0, sizeof(double4) * nl, c->stream));
  HIP_TRY_C(hipMemsetAsync(c->cams4.p, 0, sizeof(double4) * 3 * nc, c->stream));
  HIP_TRY_C(hipMemsetAsync(c->y.p, 0, sizeof(double) * 12 * nc, c->stream));
  HIP_TRY_C(hipMemsetAsync(c->q4.p, 0, sizeof(double4) * ns, c->stream));
  HIP_TRY_C(hipMemsetAsync(c->sw.p, 0, sizeof(double) * ns, c->stream));
  HIP_TRY_C(hipMemsetAsync(c->rres.p, 0, sizeof(double4) * ns, c->stream));
  // every initialisation above (uploads on the null stream, memsets on the context's non-blocking
  // stream) is complete before the context is handed out
  HIP_TRY_C(hipDeviceSynchronize());

  Dp& d = c->d;
  d.n_cams = n_cams; d.n_lms = n_lms; d.n_bins = c->n_bins; d.n_items = c->n_items;
  d.n_long = c->n_long; d.n_reg_blocks = c->n_reg_blocks;
  d.uv = c->uv.p; d.cam = c->cam.p; d.lm = c->lm.p; d.meta = c->meta.p;
  d.long_lm = c->long_lm.p; d.long_first = c->long_first.p; d.long_cnt = c->long_cnt.p;
  d.cm_slot = c->cm_slot.p; d.cm_lm = c->cm_lm.p; d.cm_uv = c->cm_uv.p;
  d.item_off = c->item_off.p; d.item_cam = c->item_cam.p; d.cam_item_off = c->cam_item_off.p;
  d.cams4 = c->cams4.p; d.cams_lin4 = c->cams_lin4.p; d.lms4 = c->lms4.p; d.lms_lin4 = c->lms_lin4.p;
  d.jl_scale4 = c->jl_scale4.p; d.hll_inv = c->hll_inv.p; d.lmrec = c->lmrec.p;
  d.cmv = CmView{c->cm_slot.p, c->cm_h.p, n_obs, c->item_off.p, c->cam_item_off.p, c->item_part.p, c->n_items, nullptr};
  d.hot_part = nullptr; d.cam_hot = c->cam_hot.p; d.n_hot_acc = c->n_hot_acc; d.n_hot_wg = c->e0c_grid;
  d.hot_rec = c->hot_rec.p;
  d.zimg = c->zimg.p;
  d.hot_cams = c->hot_cams.p; d.n_hot = std::min(n_cams, HOT_MAX);
  d.part_range = nullptr;
  d.p2p_peer = nullptr; d.p2p_epoch = nullptr; d.p2p_world = 1; d.p2p_rank = 0;
  d.sw = c->sw.p; d.rres = c->rres.p; d.q4 = c->q4.p; d.q4c = nullptr; d.cold_pos = nullptr; d.long_in_kernel = 0; d.tiles = nullptr;
  d.sigma = c->sigma.p; d.diag2 = c->diag2.p; d.G = c->G.p; d.binv = c->binv.p; d.b = c->b.p;
  d.tmp = c->tmp.p; d.accum = c->accum.p; d.z = c->z.p; d.y = c->y.p; d.inc = c->inc.p;
  d.item_part = c->item_part.p; d.item_partG = c->item_partG.p; d.cm_h = c->cm_h.p; d.n_obs = n_obs;
  d.flags = c->flags.p; d.norm_part = c->norm_part.p; d.norms = c->norms.p;
  d.sa = 0; d.sb = 1; d.eps = options->jacobi_scaling_eps; d.huber = options->huber_parameter;
  d.lambda_lm = 0; d.robust = options->robust_norm; d.scale_jl = 1;
  lap("allocations, memsets, sync");
  c->create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_create).count();
  if (timing) std::fprintf(stderr, "[povar_create] total %.1f ms\n", c->create_ms);
  *out = c;
  return 0;
}

#undef HIP_TRY_C
void povar_destroy(povar_ctx* c) {
  if (!c) return;
  c->placer_cancel.store(true);
  if (c->placer.joinable()) c->placer.join();
  (void)hipSetDevice(c->opt.device);
  c->pl_uv.release(); c->pl_cw.release(); c->pl_cpos.release(); c->pl_lm_pos.release(); c->pl_lm_of.release(); c->pl_of_slot.release();
  c->pl_c3_src.release(); c->c3_src.release();
  c->ck.release(); c->pl_ck.release(); c->ckh.release(); c->pl_ckh.release(); c->ck_zero_range.release();
  c->res.release();
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->series_graph) (void)hipGraphExecDestroy(c->series_graph);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  for (size_t p = 0; p < c->peer_host.size(); ++p)
    if (c->peer_host[p] && c->peer_host[p] != c->xbuf) (void)hipIpcCloseMemHandle(c->peer_host[p]);
  if (c->xbuf) (void)hipFree(c->xbuf);
  c->peer_dev.release(); c->p2p_epoch.release();
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->tev) (void)hipEventDestroy(e);
  c->lm_slot0.release(); c->lm_cnt_dev.release();
  c->uv.release(); c->cm_uv.release(); c->tiles.release();
  c->cam.release(); c->lm.release(); c->meta.release(); c->long_lm.release(); c->long_first.release();
  c->long_cnt.release(); c->cm_slot.release(); c->cm_lm.release(); c->item_off.release();
  c->item_cam.release(); c->cam_item_off.release(); c->flags.release();
  c->cams4.release(); c->cams_lin4.release(); c->cams_bak4.release(); c->lms4.release();
  c->lms_lin4.release(); c->lms_bak4.release(); c->jl_scale4.release(); c->rres.release(); c->q4.release();
  c->hll_inv.release(); c->sw.release(); c->sigma.release(); c->diag2.release(); c->G.release();
  c->binv.release(); c->b.release(); c->tmp.release(); c->accum.release(); c->z.release(); c->y.release();
  c->inc.release(); c->item_part.release(); c->item_partG.release(); c->norm_part.release();
  c->norms.release(); c->part.release(); c->scal.release(); c->stage.release(); c->cm_h.release(); c->lmrec.release(); c->ncw.release(); c->cc_h.release(); c->cc_part.release(); c->hot_part.release(); c->hot_rec.release(); c->zimg.release();
  c->sc_dense.release(); c->sc_xpad.release(); c->sc_lm_slot0.release(); c->sc_lm_cnt.release(); c->sc_info.release();
  c->sc_dm_part.release(); c->sc_dm.release(); c->sc_bmat.release(); c->sc_minv.release(); c->sc_x.release();
  c->sc_r.release(); c->sc_p.release(); c->sc_q.release(); c->sc_zv.release(); c->sc_part.release(); c->sc_s.release();
  c->cc_cam_range.release(); c->cold_pos.release(); c->q4c.release();
  c->v2_uv.release(); c->v2_cw.release(); c->v2_cpos.release(); c->v2_lm_pos.release(); c->v2_of_slot.release();
  c->v2_lm_of.release(); c->v2_seg.release(); c->v2_tile.release(); c->v2_wg_tile_off.release(); c->v2_wg_cam_off.release(); c->v2_wg_cams.release();
  c->v2_wg_slot_rec.release(); c->c3_lm.release(); c->v2_part_range.release(); c->c3_range.release(); c->c3_h.release(); c->v2_part.release(); c->v2_w.release(); c->v2_lmrec.release(); c->v2_lmx.release(); c->v2_lml.release(); c->v2_lsc.release();
  c->c2_lm.release(); c->c2_pos.release(); c->c2_range.release(); c->c2_h.release();
  c->cam_hot.release(); c->cc_slot.release(); c->cc_lm.release(); c->cc_item_off.release(); c->cc_cam_item_off.release(); c->hot_cams.release();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int64_t povar_device_bytes(povar_ctx* c) { return c ? (int64_t)c->bytes : 0; }

int povar_synchronize(povar_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int povar_timings_enable(povar_ctx* c, int32_t enable) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->timings_on = enable != 0;
  c->tev_used = 0;
  c->tsum = povar_timings_info{};
  return 0;
}

int povar_timings(povar_ctx* c, povar_timings_info* out) {
  if (int rc = check_ctx(c)) return rc;
  if (!out) return fail(-1, "null argument");
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (size_t i = 0; i + 1 < c->tev_used; i += 2) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->tev[i], c->tev[i + 1]));
    double* slot[5] = {&c->tsum.linearize_ms, &c->tsum.prepare_ms, &c->tsum.solve_ms, &c->tsum.apply_ms, &c->tsum.other_ms};
    int64_t* cnt[5] = {&c->tsum.linearize_calls, &c->tsum.prepare_calls, &c->tsum.solve_calls, &c->tsum.apply_calls, &c->tsum.other_calls};
    const int k = c->tev_kind[i];
    if (k >= 0 && k < 5) { *slot[k] += ms; ++*cnt[k]; }
  }
  c->tev_used = 0;
  *out = c->tsum;
  return 0;
}

int povar_get_layout_info(povar_ctx* c, povar_layout_info* out) {
  if (!c || !out) return fail(-1, "null argument");
  out->grid = c->e0c_grid;
  out->lds_slots = c->v2_max_slots;
  out->n_global = c->v2_n_global;
  out->n_tail = c->v2_n_tail;
  out->n_tiles = c->d.v2.n_tiles;
  out->n_rows = c->v2_rows;
  out->n_cold = c->n_cold3;
  out->n_obs = c->n_obs;
  out->lane_per_landmark = c->use_lpl ? 1 : 0;
  out->create_ms = c->create_ms;
  out->strategy = c->v2_strategy;
  out->hubs = c->d.v2.hubs;
  out->placement = c->placement;
  out->placement_ms = c->placer_state.load(std::memory_order_acquire) >= 2 ? c->placement_ms : 0.0;
  out->e0_kernel = c->ck_variant > 0 && c->ck.ready && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC && ck_variant_fits(c, c->ck_variant) ? c->ck_variant : 0;
  if (c->deterministic) out->e0_kernel = ck_det_possible(c) ? CK_VARIANTS + 1 : 0;  // (the fixed-point form of e0_ck)
  out->ck_ready = c->ck.ready ? 1 : 0;
  out->ck_batches = c->ck.nb;
  out->ck_slots = c->ck.slots;
  out->ck_tiles_max = c->ck.max_tiles_bt;
  out->ck_rows = c->ck.rows;
  out->ck_chunks = c->ck.n_chunks;
  out->ck_cold_chunks = c->ck.n_cold_chunks;
  out->ck_part_rec = c->ck.n_part_rec;
  out->ck_build_ms = c->ck.build_ms;
  out->e0_auto = c->ck_auto ? (c->ck_tuned ? 2 : 1) : 0;
  out->tune_lpl_us = c->ck_tune_us[0];
  out->tune_ck_us = c->ck_tune_us[1];
  out->e0_kernel_h = c->ckh_variant > 0 && c->ckh.ready && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC &&
                     c->ckh.slots <= c->ckh.stride ? 1 : 0;
  if (c->deterministic) out->e0_kernel_h = ckh_det_possible(c) ? 2 : 0;  // (2: e0_ck_h_det)
  out->ckh_ready = c->ckh.ready ? 1 : 0;
  out->ckh_batches = c->ckh.nb;
  out->ckh_slots = c->ckh.slots;
  out->ckh_chunks = c->ckh.n_chunks;
  out->ckh_cold_chunks = c->ckh.n_cold_chunks;
  out->e0_auto_h = c->ck_auto ? (c->ckh_tuned ? 2 : 1) : 0;
  out->tune_lpl_h_us = c->ckh_tune_us[0];
  out->tune_ck_h_us = c->ckh_tune_us[1];
  out->res_ready = c->res.ready ? 1 : 0;
  out->res_active = res_active(c) ? 1 : 0;
  out->res_auto = c->res_mode < 0 ? (c->res_tuned ? 2 : 1) : 0;
  out->res_wgs = c->res.W;
  out->res_waves = c->res.NW;
  out->res_rows = c->res.H;
  out->res_rounds = c->res.R;
  out->res_max_oq = c->res.max_oq;
  out->res_records = c->res.n_rec;
  out->res_max_cams = c->res.max_cam;
  out->res_max_lms = c->res.max_lm;
  out->res_max_chunks = c->res.max_chunks;
  out->res_order = c->res.order;
  out->res_lds_bytes = (int32_t)c->res.lds_bytes;
  out->res_build_ms = c->res.build_ms;
  out->tune_terms_us = c->res_tune_us[0];
  out->tune_res_us = c->res_tune_us[1];
  out->res_failed = c->res_failed ? 1 : 0;
  out->ck_packed = c->ck.ready && c->ck.packed ? 1 : 0;
  out->ck_cold_q = c->ck.ready && c->ck.cold_q ? 1 : 0;
  out->ckh_stride = c->ckh.ready ? c->ckh.stride : 0;
  out->ckh_accumulators = c->ckh.ready ? c->ckh.max_acc : 0;
  out->ckh_capped_obs = c->ckh.ready ? c->ckh.n_capped_obs : 0;
  return 0;
}

int povar_layout_finalize(povar_ctx* c, int32_t wait) {
  if (int rc = check_ctx(c)) return rc;
  if (c->placement == 2) {
    const int rc = swap_in_placed_rows(c, wait != 0);
    if (rc < 0) return rc;
    if (rc == 1) {
      // what the current linearisation left in the old row order is gone with it
      c->linearized = c->linearized_h = false;
      c->err_memo.valid = false;
    }
  }
  return c->placement == 1 || c->placement == 3 ? 1 : 0;
}

}  // extern "C"
