// povar_series.hip -- the term loop: E0 / B^-1 launchers, kernel choice by timing, hipGraph capture, the power-series entry points.
#include "povar_ctx.hpp"

CkP ck_params(const povar_ctx* c, const povar_ctx::CkDev& D) {
  return CkP{D.packed ? reinterpret_cast<const double2*>(D.uvp.p) : D.uv.p, D.li.p, D.w.p, D.tile.p, D.lane_meta.p, D.bt_off.p, D.slot_rec.p,
             D.nb, D.slots, (unsigned)(D.src.n * (D.packed ? sizeof(int2) : sizeof(double2))), (unsigned)(D.li.n * sizeof(uint32_t)),
             D.lcnt.p, D.tick.p, D.max_acc, D.packed ? 1 : 0, D.cold_q ? D.cpos.p : nullptr, c->q4c.p};
}

CkP ck_params(const povar_ctx* c) { return ck_params(c, c->ck); }


// (POVAR_DETERMINISTIC: the context is in the gather mode -- its linearisation and preparation kernels have no atomics --
// and step 1's terms run the fixed-point form of e0_ck on the records those kernels leave in lane order)
bool ck_det_possible(const povar_ctx* c) {
  return c->det_ck && c->ck.ready && c->ck.lcnt.p && c->ck.tick.p && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT &&
         c->ck.nb >= 1 && ck_lds_bytes_det(c->ck.slots, c->ck.max_acc) <= (size_t)CK_LDS_BYTES;
}

bool ck_det_active(const povar_ctx* c) { return ck_det_possible(c) && !c->joint; }

bool ck_active(const povar_ctx* c) {
  if (c->deterministic) return ck_det_active(c);
  return c->ck_variant > 0 && c->ck.ready && c->use_lpl && !c->joint && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC &&
         ck_variant_fits(c, c->ck_variant);
}

bool ckh_det_possible(const povar_ctx* c) {
  return c->det_ck && c->ckh.ready && c->ckh.lcnt.p && c->ckh.tick.p && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT &&
         c->ckh.stride == CKH_STRIDE && c->ckh.slots <= CKH_STRIDE && ckh_lds_bytes_det(c->ckh.max_acc) <= (size_t)CK_LDS_BYTES;
}

bool ckh_active(const povar_ctx* c) {
  if (c->deterministic) return c->joint && ckh_det_possible(c);
  return c->joint && c->ckh_variant > 0 && c->ckh.ready && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC &&
         (c->ckh.stride == CKH_STRIDE || c->ckh.stride == CKH_STRIDE_WIDE) && c->ckh.slots <= c->ckh.stride &&
         ckh_lds_bytes(c->ckh.max_acc, c->ckh.stride) <= (size_t)CK_LDS_BYTES;
}

// the per-camera kernels behind e0_ck / e0_ck_h: partial records only (its own table), no per-observation cold view
void ck_dp(const povar_ctx* c, Dp& da) {
  const povar_ctx::CkDev& D = c->joint ? c->ckh : c->ck;
  da.hot_part = D.part.p;
  da.part_range = D.part_range.p;
  da.cmv.src = nullptr;
  da.q_rows = 0;
  if (D.cold_q) return;  // e0_ck's cold lanes left q in the lane-per-landmark layout's cold view (camera-major, at their own
                         // positions): the per-camera kernel walks it as it does behind e0_lpl (ldsacc_dp has set h, n, cam_range, q4c)
  da.cmv.cam_range = c->ck_zero_range.p;
  da.cmv.n = 0;
}

template <int NW, int SD, bool DB, int NG>
void launch_e0_ck_t(povar_ctx* c, const Dp& da) {
  const CkP k = ck_params(c);
  const size_t lds = ck_lds_bytes(c->ck.slots, c->ck.max_acc, NG);
  const bool huber = c->opt.robust_norm == POVAR_NORM_HUBER;  // (the kernel recomputes the weights; CAUCHY's are 1: compute_error_weight)
  if (c->ck.packed) {
    if (huber) hipLaunchKernelGGL((e0_ck<NW, SD, DB, NG, true, true>), dim3(c->e0c_grid), dim3(NW * 64), lds, c->stream, da, k, c->ck.part.p);
    else hipLaunchKernelGGL((e0_ck<NW, SD, DB, NG, false, true>), dim3(c->e0c_grid), dim3(NW * 64), lds, c->stream, da, k, c->ck.part.p);
  } else {
    if (huber) hipLaunchKernelGGL((e0_ck<NW, SD, DB, NG, true, false>), dim3(c->e0c_grid), dim3(NW * 64), lds, c->stream, da, k, c->ck.part.p);
    else hipLaunchKernelGGL((e0_ck<NW, SD, DB, NG, false, false>), dim3(c->e0c_grid), dim3(NW * 64), lds, c->stream, da, k, c->ck.part.p);
  }
}

// an instantiation runs a layout whose batches fit its groups: the LDS holds ng batches at once
bool ck_variant_fits(const povar_ctx* c, int variant) {
  const CkVariant v = ck_variant_info(variant);
  return c->ck.ready && c->ck.nb % v.ng == 0 && ck_lds_bytes(c->ck.slots, c->ck.max_acc, v.ng) <= (size_t)CK_LDS_BYTES;
}

void launch_e0_ck(povar_ctx* c, const Dp& da) {
  if (c->deterministic) {  // the bit-reproducible form (povar_kernels_ck_det.hpp)
    const CkP k = ck_params(c);
    const size_t lds = ck_lds_bytes_det(c->ck.slots, c->ck.max_acc);
    if (c->opt.robust_norm == POVAR_NORM_HUBER)
      hipLaunchKernelGGL((e0_ck_det<16, 2, true>), dim3(c->e0c_grid), dim3(1024), lds, c->stream, da, k, c->ck.part.p);
    else
      hipLaunchKernelGGL((e0_ck_det<16, 2, false>), dim3(c->e0c_grid), dim3(1024), lds, c->stream, da, k, c->ck.part.p);
    return;
  }
  switch (c->ck_variant) {
#define X(id, nw, sd, db, ng) case id: launch_e0_ck_t<nw, sd, db, ng>(c, da); break;
    POVAR_CK_VARIANTS(X)
#undef X
    default: break;
  }
}

template <int NW, int SD, bool DB, int NG>
hipError_t ck_set_lds_t() {
  hipError_t e = hipFuncSetAttribute((const void*)e0_ck<NW, SD, DB, NG, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)e0_ck<NW, SD, DB, NG, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)e0_ck<NW, SD, DB, NG, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)e0_ck<NW, SD, DB, NG, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
  return e;
}

hipError_t ck_set_lds_all() {
  hipError_t e = hipFuncSetAttribute((const void*)e0_ck_det<16, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)e0_ck_h_det<16, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)e0_ck_h_det<16, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)e0_ck_det<16, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES);
#define X(id, nw, sd, db, ng) if (e == hipSuccess) e = ck_set_lds_t<nw, sd, db, ng>();
  POVAR_CK_VARIANTS(X)
#undef X
  return e;
}

// Which of the two step-1 E0 kernels is faster depends on the graph (e0_ck: venice-like camera counts, any share of
// observations whose camera has no LDS slot; e0_lpl: many cameras and few observations per (camera, batch), where a chunk
// is a single observation -- final-13682).  Unless the caller has forced one, both are timed once per layout on the
// problem itself: a warm-up and three launches each on the prepared system (they only write their partial records).


// robust weights in chunk order (V2::w is written by the linearisation walk in lane-per-landmark order)
void ensure_ck_w(povar_ctx* c) {
  povar_ctx::CkDev& D = c->joint ? c->ckh : c->ck;
  if (!c->opt.robust_norm || !D.ready || !D.w.p || !c->v2_w.p || D.w_lin_id == c->lin_id) return;
  const int64_t n = (int64_t)D.src.n;  // every row, the padding included (weight 0)
  hipLaunchKernelGGL(ck_gather_w, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, (const int*)D.src.p, (const double*)c->v2_w.p,
                     D.w.p, n);
  D.w_lin_id = c->lin_id;
}

void launch_e0_ck_h(povar_ctx* c, const Dp& da) {
  const CkP k = ck_params(c, c->ckh);
  if (c->deterministic) {  // the bit-reproducible form (povar_kernels_ck_det.hpp)
    const size_t ldsd = ckh_lds_bytes_det(c->ckh.max_acc);
    if (c->opt.robust_norm)
      hipLaunchKernelGGL((e0_ck_h_det<16, 2, true>), dim3(c->e0c_grid), dim3(1024), ldsd, c->stream, da, k, c->ckh.part.p);
    else
      hipLaunchKernelGGL((e0_ck_h_det<16, 2, false>), dim3(c->e0c_grid), dim3(1024), ldsd, c->stream, da, k, c->ckh.part.p);
    return;
  }
  const size_t lds = ckh_lds_bytes(c->ckh.max_acc, c->ckh.stride);
  if (c->ckh.stride == CKH_STRIDE_WIDE) {  // two landmark batches instead of three, fewer accumulators (povar_kernels_ck_joint.hpp)
    if (c->opt.robust_norm)
      hipLaunchKernelGGL((e0_ck_h<16, 2, true, CKH_STRIDE_WIDE>), dim3(c->e0c_grid), dim3(1024), lds, c->stream, da, k, c->ckh.part.p);
    else
      hipLaunchKernelGGL((e0_ck_h<16, 2, false, CKH_STRIDE_WIDE>), dim3(c->e0c_grid), dim3(1024), lds, c->stream, da, k, c->ckh.part.p);
    return;
  }
  if (c->opt.robust_norm)
    hipLaunchKernelGGL((e0_ck_h<16, 2, true>), dim3(c->e0c_grid), dim3(1024), lds, c->stream, da, k, c->ckh.part.p);
  else
    hipLaunchKernelGGL((e0_ck_h<16, 2, false>), dim3(c->e0c_grid), dim3(1024), lds, c->stream, da, k, c->ckh.part.p);
}


// instantiations: wavefronts per workgroup, rows per chunk, chunks per lane, landmark slots per lane.  1024-thread
// workgroups (128 VGPRs per lane): one chunk of at most two rows; 512-thread ones (256): two chunks of up to four rows
#define POVAR_RES_VARIANTS(X) X(16, 1, 1, 1) X(16, 2, 1, 1) X(8, 1, 2, 1) X(8, 1, 2, 2) X(8, 2, 2, 1) X(8, 2, 2, 2) X(8, 4, 2, 1) X(8, 4, 2, 2)
template <int NW, int H, int RR, int LS>
void launch_res_t(povar_ctx* c, const ResP& k) {
  if (c->opt.robust_norm)
    hipLaunchKernelGGL((series_res<NW, H, RR, LS, true>), dim3(c->res.W), dim3(NW * 64), c->res.lds_bytes, c->stream, c->d, k);
  else
    hipLaunchKernelGGL((series_res<NW, H, RR, LS, false>), dim3(c->res.W), dim3(NW * 64), c->res.lds_bytes, c->stream, c->d, k);
}

bool res_variant_exists(int nw, int h, int rr, int ls) {
#define X(NW_, H_, R_, LS_) if (nw == NW_ && h == H_ && rr == R_ && ls == LS_) return true;
  POVAR_RES_VARIANTS(X)
#undef X
  return false;
}

void launch_res(povar_ctx* c, const ResP& k) {
#define X(NW_, H_, R_, LS_)                                                              \
  if (c->res.NW == NW_ && c->res.H == H_ && c->res.R == R_ && c->res.LS == LS_) {        \
    launch_res_t<NW_, H_, R_, LS_>(c, k);                                                \
    return;                                                                              \
  }
  POVAR_RES_VARIANTS(X)
#undef X
}

template <int NW, int H, int RR, int LS>
hipError_t res_set_lds_t() {
  hipError_t e = hipFuncSetAttribute((const void*)series_res<NW, H, RR, LS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_BYTES);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)series_res<NW, H, RR, LS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_BYTES);
}

hipError_t res_set_lds_all() {
  hipError_t e = hipSuccess;
#define X(NW_, H_, R_, LS_) if (e == hipSuccess) e = res_set_lds_t<NW_, H_, R_, LS_>();
  POVAR_RES_VARIANTS(X)
#undef X
  return e;
}

// the context can run the resident series now (whether it SHOULD is res_mode / the timing of res_autotune)
bool res_possible(const povar_ctx* c) {
  // (a communicator of ONE rank exchanges nothing: such a context -- the one-GPU proxy of a shard, tools/shard_sweep.sh --
  // is as good as unsharded; with peers the resident kernel would need their sums inside the launch: not built)
  return c->res.ready && !c->res_failed && !c->joint && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC && !c->profile &&
         (!sharded(c) || (c->world == 1 && !c->p2p));
}

bool res_active(const povar_ctx* c) {
  return res_possible(c) && (c->res_mode == 1 || (c->res_mode < 0 && c->res_tuned && c->res_choice));  // (and m <= 250: run_series' caller)
}

ResP res_params(const povar_ctx* c, int m, double q_tol, double r_tol) {
  const povar_ctx::ResDev& D = c->res;
  ResP k{};
  k.lane_cam = D.lane_cam.p; k.lane_seg = D.lane_seg.p;
  k.uv = D.uv.p; k.lslot = D.lslot.p; k.oslot = D.oslot.p; k.wave_h = D.wave_h.p;
  k.lm_off = D.lm_off.p; k.lm_id = D.lm_id.p; k.cam_off = D.cam_off.p; k.cam_id = D.cam_id.p; k.cam_zi = D.cam_zi.p;
  k.own_off = D.own_off.p; k.own_cam = D.own_cam.p; k.own_zi = D.own_zi.p; k.own_q = D.own_q.p;
  k.oq_off = D.oq_off.p; k.oq_rec = D.oq_rec.p;
  k.part = D.part.p; k.zbuf = D.zbuf.p; k.nrm = D.nrm.p; k.launch = D.launch.p;
  k.part_bytes = (unsigned)(D.part.n * sizeof(uint4)); k.z_bytes = (unsigned)(D.zbuf.n * sizeof(uint4)); k.nrm_bytes = (unsigned)(D.nrm.n * sizeof(uint4));
  k.W = D.W; k.m = m;
  k.want_norms = (q_tol > 0 || r_tol > 0) ? 1 : 0;
  k.want_norm0 = r_tol > 0 ? 1 : 0;
  // the robust weights of the linearisation in force: per slot (sqrt, lane-per-observation linearisation) or in row order
  k.w_mode = c->aux_lin_id == c->lin_id ? 2 : 1;
  k.q_tol = q_tol; k.r_tol = r_tol;
  k.spin_limit = c->res_spin_limit;
  return k;
}

// the whole series as one launch (+ the node that numbers the next one)
int enqueue_series_res(povar_ctx* c, int32_t m, double q_tol, double r_tol) {
  HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));
  launch_res(c, res_params(c, m, q_tol, r_tol));
  hipLaunchKernelGGL(res_bump_launch, dim3(1), dim3(1), 0, c->stream, c->res.launch.p);
  return 0;
}

void prof_mark(povar_ctx* c, int kind) {
  if (!c->profile) return;
  if (c->ev_used == c->ev.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    c->ev.push_back(e);
    c->ev_kind.push_back(-1);
  }
  c->ev_kind[c->ev_used] = kind;
  (void)hipEventRecord(c->ev[c->ev_used], c->stream);
  ++c->ev_used;
}

// Dp of the per-term kernels in POVAR_E0_IMPLICIT_LDSACC mode: cold camera-major view + hot partials
Dp ldsacc_dp(povar_ctx* c, bool long_in_kernel) {
  Dp dt = c->d;
  dt.cmv = CmView{c->cc_slot.p, c->cc_h.p, c->n_cold, c->cc_item_off.p, c->cc_cam_item_off.p, c->cc_part.p,
                  c->n_cold_items, c->cc_cam_range.p};
  dt.hot_part = c->hot_part.p;
  dt.q4c = c->q4c.p;
  dt.cold_pos = c->cold_pos.p;
  if (c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC) {
    // e0_lpl: its own cold view (observations whose camera is not resident in their workgroup) and partial records
    dt.cmv.h = c->c3_h.p;
    dt.cmv.n = c->n_cold3;
    dt.cmv.cam_range = c->c3_range.p;
    dt.cmv.src = c->q_rows ? c->c3_src.p : nullptr;
    dt.q_rows = c->q_rows ? 1 : 0;
    dt.hot_part = c->v2_part.p;
    dt.part_range = c->v2_part_range.p;
    dt.cold_pos = nullptr;
    dt.long_in_kernel = 1;
    return dt;
  }
  if (long_in_kernel && c->long_in_kernel) {
    // view "A": e0_lm_cached<true> walks the long landmarks itself, their LDS-accumulated observations are not cold
    dt.cmv.h = c->c2_h.p;
    dt.cmv.n = c->n_cold2;
    dt.cmv.cam_range = c->c2_range.p;
    dt.cold_pos = c->c2_pos.p;
    dt.long_in_kernel = 1;
  }
  return dt;
}

// E0 x for the current term: implicit (LM pass, CM pass) or stored tiles.  The per-camera
// result is consumed by cam_binv_axpy (mode 1: scatter items, mode 2: dense y).
// fuse_norms >= 0: the caller is the term loop and takes B^-1 + AXPY next with want_norms = fuse_norms,
// so the unsharded step-1 LDSACC path may run them inside the per-camera sum (binv_mode 4: done)
int launch_e0(povar_ctx* c, int* binv_mode, int fuse_norms) {
  prof_mark(c, 0);
  if (!(c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC)) ensure_legacy(c);  // cm_scatter / legacy cold views
  if (ck_active(c) || ckh_active(c)) ensure_ck_w(c);  // (a no-op inside the graph capture: the solve entry points have called it before)
  if (c->joint) {
    const bool ckh_now = ckh_active(c);
    const bool acc = c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC || ckh_now;  // (e0_ck_h_det leaves partial records too)
    Dp dj = ldsacc_dp(c, true);  // what the per-camera kernels below see: e0_ck_h leaves partial records only
    if (ckh_now) ck_dp(c, dj);
    if (ckh_now) {
      launch_e0_ck_h(c, dj);
    } else if (acc && c->use_lpl) {
      Dp da = ldsacc_dp(c, true);
      if (c->opt.robust_norm)
        hipLaunchKernelGGL(e0_lpl_h<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK), lpl_lds_bytes_h(c->v2_max_slots), c->stream, da, c->v2_part.p);
      else
        hipLaunchKernelGGL(e0_lpl_h<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK), lpl_lds_bytes_h(c->v2_max_slots), c->stream, da, c->v2_part.p);
    } else if (acc) {
      // cold observations write q to their camera-major position (q4c); long landmarks are walked inside the kernel
      const Dp da = ldsacc_dp(c, true);
      hipLaunchKernelGGL(e0_lm_cached_h, dim3(c->e0c_grid), dim3(E0C_BLOCK),
                         (size_t)c->n_hot_acc * (HOT_REC_H * sizeof(double2) + 96), c->stream, da,
                         c->e0c_bins_per_wg, c->hot_part.p);
      if (c->n_long > 0 && !c->long_in_kernel)
        hipLaunchKernelGGL((lm_long<OpE0H>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, da, OpE0H{}, c->part.p);
    } else {
      launch_lm(c, OpE0H{});
    }
    if (acc && fuse_norms >= 0 && !sharded(c) && c->fuse_binv) {
      hipLaunchKernelGGL(cam_cold_sum_binv_h<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, dj, fuse_norms,
                         (const double*)c->ncw.p);
      *binv_mode = 4;  // B^-1, AXPY and z already done
    } else if (acc) {
      hipLaunchKernelGGL(cam_cold_sum<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, dj, 1);
      *binv_mode = 2;  // dense y (sigma applied)
    } else {
      hipLaunchKernelGGL(cm_scatter, dim3(grid_for(std::max(c->n_items, 1), 4)), dim3(256), 0, c->stream, c->d, 1, 1);
      *binv_mode = 1;
      if (sharded(c)) {
        hipLaunchKernelGGL(cam_sum_items, dim3(grid_for(c->n_cams, 4)), dim3(256), 0, c->stream, c->d, c->d.y, 1);
        *binv_mode = 2;
      }
    }
  } else {
    const bool ck_now = ck_active(c);
    // (the fixed-point e0_ck of the deterministic mode leaves partial records like the LDS-accumulating kernels)
    const bool acc = c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC || c->opt.e0_mode == POVAR_E0_TILES_LDSACC || ck_now;
    // ACC: cold observations write q to their camera-major position (q4c); the implicit form also walks the
    // long landmarks inside e0_lm_cached (its own cold view)
    const bool lik = (c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC && c->long_in_kernel) || ck_now;
    Dp da = acc ? ldsacc_dp(c, lik) : c->d;
    if (ck_now) {
      ck_dp(c, da);
      da.long_in_kernel = 1;  // (e0_ck walks every landmark)
    }
    // peer-to-peer exchange: only inside the term loop (fuse_norms >= 0) of the lane-per-landmark kernels; every other
    // caller (right_mul_e0, PCG) wants the dense, all-reduced y
    const bool p2p_now = c->p2p && fuse_norms >= 0 && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC;
    if (p2p_now) p2p_dp(c, da);
    if (c->opt.e0_mode == POVAR_E0_TILES) launch_lm(c, OpE0Tiles{});
    else if (c->opt.e0_mode == POVAR_E0_TILES_LDSACC) {
      hipLaunchKernelGGL(e0_tiles_cached, dim3(c->e0c_grid), dim3(E0T_BLOCK),
                         (size_t)c->n_hot_acc * (HOT_REC_T * sizeof(double2) + 96), c->stream, da,
                         c->e0c_bins_per_wg, c->hot_part.p);
      if (c->n_long > 0)
        hipLaunchKernelGGL((lm_long<OpE0Tiles>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, da, OpE0Tiles{}, c->part.p);
    }
    else if (ck_now)
      launch_e0_ck(c, da);
    else if (c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC && c->use_lpl && c->opt.robust_norm)
      hipLaunchKernelGGL(e0_lpl<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK),
                         lpl_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    else if (c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC && c->use_lpl)
      hipLaunchKernelGGL(e0_lpl<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK),
                         lpl_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    else if (c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC)
      hipLaunchKernelGGL(e0_lm_cached<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK),
                         (size_t)c->n_hot_acc * (HOT_REC * sizeof(double2) + 96), c->stream, da,
                         c->e0c_bins_per_wg, c->hot_part.p);
    else
      hipLaunchKernelGGL(e0_lm_cached<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK),
                         (size_t)c->n_hot * HOT_REC * sizeof(double2), c->stream, c->d, c->e0c_bins_per_wg,
                         (double*)nullptr);
    if ((c->opt.e0_mode == POVAR_E0_IMPLICIT || c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC) && c->n_long > 0 && !lik)
      hipLaunchKernelGGL((lm_long<OpE0>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, da, OpE0{}, c->part.p);
    if (acc && fuse_norms >= 0 && !sharded(c) && c->fuse_binv) {
      // 128 threads per camera: a camera's run is at most one partial record per workgroup (256) plus ~85 cold
      // observations; two wavefronts keep four loads per thread in flight and halve the cross-wavefront reduction
      // (256 threads: 71.7 us per term, 128: 69.5, 64: 69.6 on venice-1778)
      hipLaunchKernelGGL(cam_cold_sum_binv<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, da, fuse_norms);
      *binv_mode = 4;  // B^-1, AXPY and z already done
    } else if (acc) {
      hipLaunchKernelGGL(cam_cold_sum<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, da, 0);
      *binv_mode = 2;  // dense y (sigma applied)
    } else {
      hipLaunchKernelGGL(cm_scatter, dim3(grid_for(std::max(c->n_items, 1), 4)), dim3(256), 0, c->stream, c->d, 1, 0);
      *binv_mode = 1;
      if (sharded(c)) {
        hipLaunchKernelGGL(cam_sum_items, dim3(grid_for(c->n_cams, 4)), dim3(256), 0, c->stream, c->d, c->d.y, 1);
        *binv_mode = 2;
      }
    }
  }
  if (c->p2p && fuse_norms >= 0 && *binv_mode == 2 && !c->joint && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC) {
    *binv_mode = 5;  // cam_cold_sum pushed the partials to the peers; cam_binv_axpy waits for the world's slabs
    return 0;
  }
  if (sharded(c)) {
    int rc = allreduce(c, c->d.y, 12 * (size_t)c->n_cams);
    if (rc) return rc;
  }
  return 0;
}

void launch_binv(povar_ctx* c, int mode, int want_norms) {
  if (mode == 4) return;  // fused into cam_cold_sum_binv
  prof_mark(c, 1);
  if (c->joint) {
    const Dp dt = mode == 3 ? ldsacc_dp(c) : c->d;
    hipLaunchKernelGGL(cam_binv_axpy_h, dim3(c->n_cam_blocks), dim3(K9_CAMS * 64), 0, c->stream, dt, mode == 3 ? 1 : mode,
                       want_norms, (const double*)c->ncw.p);
  }
  else {
    Dp dt = mode == 3 ? ldsacc_dp(c) : c->d;  // 3: item sums over the cold view + LDS partials
    if (mode == 5) p2p_dp(c, dt);
    hipLaunchKernelGGL(cam_binv_axpy, dim3(c->n_cam_blocks), dim3(K9_CAMS * 64), 0, c->stream, dt, mode == 3 ? 1 : mode,
                       want_norms);
  }
}

int ck_autotune(povar_ctx* c) {
  if (!c->ck_auto || c->ck_tuned || !c->ck.ready || !c->use_lpl || c->joint || c->opt.e0_mode != POVAR_E0_IMPLICIT_LDSACC ||
      !ck_variant_fits(c, 1))
    return 0;
  c->ck_tuned = true;
  ensure_ck_w(c);
  HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));  // (a series that ended early leaves "done" set)
  // Two rounds of (warm-up + REPS launches) of each kernel, alternating, the FASTER round of each counts: one round's mean
  // was seen 18 % off on the same box (Zipf(0.5): e0_ck 80.2 against 67.9 us in two processes -- clocks still ramping, the
  // placement thread's uploads), enough to keep the slower kernel for the life of the layout.
  constexpr int ROUNDS = 2;
  EventSet<4 * ROUNDS> ev;
  HIP_TRY(ev.create());
  Dp da = ldsacc_dp(c, true);
  da.p2p_peer = nullptr;
  da.p2p_epoch = nullptr;
  Dp dk = da;
  ck_dp(c, dk);
  const int keep = c->ck_variant;
  struct Restore {  // a failure below leaves the choice as it was and the timing to be repeated
    povar_ctx* c; int keep; bool done = false;
    ~Restore() { if (!done) { c->ck_variant = keep; c->ck_tuned = false; } }
  } restore{c, keep};
  c->ck_variant = 1;
  // What is timed is the PAIR of a term -- the E0 kernel and the per-camera kernel behind it --: e0_lpl leaves its cold
  // observations to the per-camera kernel (Zipf(0.5): 25 us there against 7 behind e0_ck; the E0 kernels alone were a draw
  // in some processes and the slower pair was kept).  The per-camera kernel's outputs (accum, tmp, z) are what the series'
  // first kernel writes anyway.
  auto run_lpl = [&]() {
    if (c->opt.robust_norm)
      hipLaunchKernelGGL(e0_lpl<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK), lpl_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    else
      hipLaunchKernelGGL(e0_lpl<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK), lpl_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    hipLaunchKernelGGL(cam_cold_sum_binv<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, da, 0);
  };
  auto run_ck = [&]() {
    launch_e0_ck(c, dk);
    hipLaunchKernelGGL(cam_cold_sum_binv<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, dk, 0);
  };
  constexpr int REPS = 3;
  for (int r = 0; r < ROUNDS; ++r) {
    run_lpl();
    HIP_TRY(hipEventRecord(ev[4 * r], c->stream));
    for (int i = 0; i < REPS; ++i) run_lpl();
    HIP_TRY(hipEventRecord(ev[4 * r + 1], c->stream));
    run_ck();
    HIP_TRY(hipEventRecord(ev[4 * r + 2], c->stream));
    for (int i = 0; i < REPS; ++i) run_ck();
    HIP_TRY(hipEventRecord(ev[4 * r + 3], c->stream));
  }
  c->ck_variant = keep;
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipGetLastError());
  float ms_lpl = 1e30f, ms_ck = 1e30f;
  for (int r = 0; r < ROUNDS; ++r) {
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, ev[4 * r], ev[4 * r + 1]));
    HIP_TRY(hipEventElapsedTime(&b, ev[4 * r + 2], ev[4 * r + 3]));
    ms_lpl = std::min(ms_lpl, a);
    ms_ck = std::min(ms_ck, b);
  }
  restore.done = true;
  c->ck_tune_us[0] = 1e3f * ms_lpl / REPS;
  c->ck_tune_us[1] = 1e3f * ms_ck / REPS;
  c->ck_variant = ms_ck < 0.98f * ms_lpl ? 1 : 0;
  c->ck_fresh[0] = true;
  return 0;
}

// The ranks of a sharded run keep the SAME term kernel (each rank timed its own shard and ranks of one run could
// end on different kernels -- a term then takes as long as the slower choice, and two runs of the same problem need not agree).
// Every rank's prepare call ends in one all-reduce of four doubles through the context's exchange (RCCL or the host hook):
// the timings of the ranks that have just timed (their sum decides, for everybody), how many did, and how many ranks have no
// chunk layout (one is enough to keep every rank on e0_lpl).  Unconditional on anything a rank decides by itself -- when a
// rank's placed rows arrive, and with them a new timing, differs from rank to rank; the collective does not.
int tune_agree(povar_ctx* c, int step) {
  // (the condition holds on every rank or on none: options and environment are the run's, not the rank's)
  if (!sharded(c) || !c->ck_auto || c->deterministic || c->opt.e0_mode != POVAR_E0_IMPLICIT_LDSACC) return 0;
  const bool fresh = c->ck_fresh[step];
  const float* us = step ? c->ckh_tune_us : c->ck_tune_us;
  const bool ready = c->use_lpl && (step ? c->ckh.ready : c->ck.ready);
  double h[4] = {fresh ? us[0] : 0.0, fresh ? us[1] : 0.0, fresh ? 1.0 : 0.0, ready ? 0.0 : 1.0};
  HIP_TRY(hipMemcpyAsync(c->scal.p, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // (h is on this frame)
  if (int rc = allreduce(c, c->scal.p, 4)) return rc;
  HIP_TRY(hipMemcpyAsync(h, c->scal.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->ck_fresh[step] = false;
  int& variant = step ? c->ckh_variant : c->ck_variant;
  if (h[3] > 0) variant = 0;
  else if (h[2] > 0) variant = h[1] < 0.98 * h[0] ? 1 : 0;
  return 0;
}

// the same choice for step 2: e0_lpl_h against e0_ck_h on the prepared joint system
int ckh_autotune(povar_ctx* c) {
  if (!c->ck_auto || c->ckh_tuned || !c->joint || !c->ckh.ready || !c->use_lpl || c->opt.e0_mode != POVAR_E0_IMPLICIT_LDSACC) return 0;
  c->ckh_tuned = true;
  c->ckh_variant = 1;
  if (!ckh_active(c)) {
    c->ckh_variant = 0;
    return 0;
  }
  ensure_ck_w(c);
  HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));
  constexpr int ROUNDS = 2;
  EventSet<4 * ROUNDS> ev;
  HIP_TRY(ev.create());
  struct Restore {
    povar_ctx* c; bool done = false;
    ~Restore() { if (!done) { c->ckh_variant = 0; c->ckh_tuned = false; } }
  } restore{c};
  Dp da = ldsacc_dp(c, true);
  Dp dk = da;
  ck_dp(c, dk);
  auto run_lpl = [&]() {
    if (c->opt.robust_norm)
      hipLaunchKernelGGL(e0_lpl_h<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK), lpl_lds_bytes_h(c->v2_max_slots), c->stream, da, c->v2_part.p);
    else
      hipLaunchKernelGGL(e0_lpl_h<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK), lpl_lds_bytes_h(c->v2_max_slots), c->stream, da, c->v2_part.p);
    hipLaunchKernelGGL(cam_cold_sum_binv_h<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, da, 0, (const double*)c->ncw.p);
  };
  auto run_ck = [&]() {
    launch_e0_ck_h(c, dk);
    hipLaunchKernelGGL(cam_cold_sum_binv_h<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, dk, 0, (const double*)c->ncw.p);
  };
  constexpr int REPS = 3;
  for (int r = 0; r < ROUNDS; ++r) {  // (two alternating rounds of the term's pair, the faster one of each counts: see ck_autotune)
    run_lpl();
    HIP_TRY(hipEventRecord(ev[4 * r], c->stream));
    for (int i = 0; i < REPS; ++i) run_lpl();
    HIP_TRY(hipEventRecord(ev[4 * r + 1], c->stream));
    run_ck();
    HIP_TRY(hipEventRecord(ev[4 * r + 2], c->stream));
    for (int i = 0; i < REPS; ++i) run_ck();
    HIP_TRY(hipEventRecord(ev[4 * r + 3], c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipGetLastError());
  float ms_lpl = 1e30f, ms_ck = 1e30f;
  for (int r = 0; r < ROUNDS; ++r) {
    float ta = 0, tb = 0;
    HIP_TRY(hipEventElapsedTime(&ta, ev[4 * r], ev[4 * r + 1]));
    HIP_TRY(hipEventElapsedTime(&tb, ev[4 * r + 2], ev[4 * r + 3]));
    ms_lpl = std::min(ms_lpl, ta);
    ms_ck = std::min(ms_ck, tb);
  }
  restore.done = true;
  c->ckh_tune_us[0] = 1e3f * ms_lpl / REPS;
  c->ckh_tune_us[1] = 1e3f * ms_ck / REPS;
  c->ckh_variant = ms_ck < 0.98f * ms_lpl ? 1 : 0;
  c->ck_fresh[1] = true;
  return 0;
}

extern "C" {


int povar_power_series_begin(povar_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;  // (a resident series that gave up is repeated BEFORE its state is overwritten / continued)
  HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));
  launch_binv(c, 0, 0);
  HIP_TRY(hipGetLastError());
  return 0;
}

int povar_power_series_step(povar_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;  // (a resident series that gave up is repeated BEFORE its state is overwritten / continued)
  int mode = 1;
  if (int rc = launch_e0(c, &mode, 0)) return rc;
  launch_binv(c, mode, 0);
  HIP_TRY(hipGetLastError());
  return 0;
}

int enqueue_series(povar_ctx* c, int32_t m, double q_tol, double r_tol) {
  const bool norms = q_tol > 0 || r_tol > 0;
  HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));
  launch_binv(c, 0, (m > 0 && r_tol > 0) ? 1 : 0);
  if (m > 0 && r_tol > 0)
    hipLaunchKernelGGL(series_check, dim3(1), dim3(64), 0, c->stream, c->d, c->n_cam_blocks, 0, q_tol, r_tol);
  for (int i = 1; i <= m; ++i) {
    int mode = 1;
    if (int rc = launch_e0(c, &mode, norms ? 1 : 0)) return rc;
    launch_binv(c, mode, norms ? 1 : 0);
    if (norms)  // the fused kernel leaves one norm partial per camera, cam_binv_axpy one per workgroup
      hipLaunchKernelGGL(series_check, dim3(1), dim3(64), 0, c->stream, c->d, mode == 4 ? c->n_cams : c->n_cam_blocks, i,
                         q_tol, r_tol);
  }
  prof_mark(c, -1);
  return 0;
}

// One series on the context's stream: the per-term kernels (use_res = false) or the resident kernel, through the cached
// hipGraph where the context allows a capture.
int run_series(povar_ctx* c, int32_t m, double q_tol, double r_tol, bool use_res) {
  const bool norms = q_tol > 0 || r_tol > 0;
  // with a communicator the loop is launched kernel by kernel (the per-term all-reduce dominates and
  // RCCL-in-capture is not something a 1-GPU box can validate); POVAR_GRAPH_COMM=1 opts in
  const bool p2p_terms = c->p2p && !c->joint && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC;
  if (c->use_graph && !c->profile && m > 0 && (use_res || p2p_terms || (!c->host_fn && (!c->comm || c->graph_with_comm)))) {
    // the whole loop (memset, B^-1, m x {E0 kernels, [all-reduce], B^-1 + AXPY, [check]}) is one graph
    // launch; it is re-captured only when a kernel argument changes
    const int key[6] = {m, (c->joint ? 1 : 0) | (use_res ? 2 : 0) | (use_res ? (res_params(c, m, q_tol, r_tol).w_mode << 2) : 0),
                        c->opt.e0_mode + 16 * (c->joint ? (ckh_active(c) ? 1 : 0) : ck_active(c) ? c->ck_variant : 0), (sharded(c) ? 1 : 0) | (p2p_terms ? 2 : 0), norms ? 1 : 0, r_tol > 0 ? 1 : 0};
    // (the landmark damping is an argument of the prepare / back-substitution kernels only: no kernel of the loop reads
    // it, and step 2 changes it with every LM iteration -- a capture + instantiation of 0.25 ms each time)
    Dp key_d = c->d;
    key_d.lambda_lm = 0;
    const bool same = c->series_graph && std::memcmp(key, c->series_graph_key, sizeof(key)) == 0 &&
                      std::memcmp(&key_d, &c->series_graph_d, sizeof(Dp)) == 0 &&
                      c->series_graph_tol[0] == q_tol && c->series_graph_tol[1] == r_tol;
    if (!same) {
      if (c->series_graph) (void)hipGraphExecDestroy(c->series_graph);
      c->series_graph = nullptr;
      hipGraph_t g = nullptr;
      std::lock_guard<std::mutex> lk(g_capture_mu);  // no HIP call of the row-placement thread inside the capture
      HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      const int rc = use_res ? enqueue_series_res(c, m, q_tol, r_tol) : enqueue_series(c, m, q_tol, r_tol);
      hipError_t e = hipStreamEndCapture(c->stream, &g);
      if (rc) return rc;
      HIP_TRY(e);
      HIP_TRY(hipGraphInstantiate(&c->series_graph, g, nullptr, nullptr, 0));
      (void)hipGraphDestroy(g);
      std::memcpy(c->series_graph_key, key, sizeof(key));
      c->series_graph_d = key_d;
      c->series_graph_tol[0] = q_tol;
      c->series_graph_tol[1] = r_tol;
    }
    HIP_TRY(hipGraphLaunch(c->series_graph, c->stream));
  } else if (use_res) {
    if (int rc = enqueue_series_res(c, m, q_tol, r_tol)) return rc;
  } else {
    if (int rc = enqueue_series(c, m, q_tol, r_tol)) return rc;
  }
  if (use_res) {
    c->res_check = true;  // the give-up bit is looked at with the caller's next read-back (res_verify)
    c->res_last_m = m;
    c->res_last_tol[0] = q_tol;
    c->res_last_tol[1] = r_tol;
    c->flag0_clean = false;
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// A resident series gives up (bit 2 of flags[0]) when its workgroups were not all on the device together -- another
// context's kernels held CUs for longer than the bounded spins.  The result is then incomplete: the series is repeated
// with the per-term kernels and the context stays on them.  Called before anything reads what the series left.
int res_verify(povar_ctx* c) {
  if (c->det_check) {
    c->det_check = false;
    int f[4];
    if (int rc = read_flags(c, f)) return rc;
    if (f[0] & 8) {
      HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(int), c->stream));
      return fail(-3, "e0_ck_det: the accumulator tickets of the chunk layout do not match the kernel's tile walk");
    }
  }
  if (!c->res_check) return 0;
  c->res_check = false;
  int f[4];
  if (int rc = read_flags(c, f)) return rc;
  if (!(f[0] & 4)) return 0;
  c->res_failed = true;
  HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(int), c->stream));
  if (c->series_graph) { (void)hipGraphExecDestroy(c->series_graph); c->series_graph = nullptr; }
  return run_series(c, c->res_last_m, c->res_last_tol[0], c->res_last_tol[1], false);
}

// Which of the two forms of the series is faster is a property of the context (observations per workgroup, cameras per
// workgroup): unless one is forced (POVAR_RES, povar_set_series_kernel) both are run once on the caller's prepared system
// -- a warm-up and REPS timed solves each, the same m and tolerances -- and the faster one is kept.
int res_autotune(povar_ctx* c, int32_t m, double q_tol, double r_tol) {
  if (c->res_mode >= 0 || c->res_tuned || !res_possible(c) || m < 4 || m > 250) return 0;
  c->res_tuned = true;
  struct Restore {  // a failure below leaves the choice open and the timing to be repeated (as ck_autotune)
    povar_ctx* c; bool done = false;
    ~Restore() { if (!done) { c->res_tuned = false; c->res_choice = false; } }
  } restore{c};
  // Two alternating rounds of (warm-up + REPS series) of each form, the FASTER round of each counts -- one round's mean was
  // seen 18 % off in some processes (ck_autotune) --, and both forms run all m terms: the tolerances are off while timing (an
  // early exit would time a few terms of one form against a few of the other; the caller's series follows with its own)
  (void)q_tol; (void)r_tol;
  constexpr int ROUNDS = 2, REPS = 2;
  EventSet<4 * ROUNDS> ev;
  HIP_TRY(ev.create());
  for (int r = 0; r < ROUNDS; ++r)
    for (int which = 0; which < 2; ++which) {
      if (int rc = run_series(c, m, 0.0, -1.0, which == 1)) return rc;
      HIP_TRY(hipEventRecord(ev[4 * r + 2 * which], c->stream));
      for (int i = 0; i < REPS; ++i)
        if (int rc = run_series(c, m, 0.0, -1.0, which == 1)) return rc;
      HIP_TRY(hipEventRecord(ev[4 * r + 2 * which + 1], c->stream));
      if (which == 1) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (int rc = res_verify(c)) return rc;
        if (c->res_failed) { restore.done = true; c->res_choice = false; return 0; }
      }
    }
  HIP_TRY(hipStreamSynchronize(c->stream));
  float ms[2] = {1e30f, 1e30f};
  for (int r = 0; r < ROUNDS; ++r)
    for (int which = 0; which < 2; ++which) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, ev[4 * r + 2 * which], ev[4 * r + 2 * which + 1]));
      ms[which] = std::min(ms[which], t);
    }
  restore.done = true;
  c->res_tune_us[0] = 1e3f * ms[0] / (REPS * m);
  c->res_tune_us[1] = 1e3f * ms[1] / (REPS * m);
  c->res_choice = ms[1] < 0.98f * ms[0];
  return 0;
}

int povar_power_series_pose(povar_ctx* c, int32_t m, double q_tol, double r_tol, int32_t* num_iterations,
                            int32_t* termination) {
  if (int rc = check_ctx(c)) return rc;
  if (m < 0) return fail(-1, "power_sc_iterations < 0");
  TimeScope ts(c, 2);
  if (!(c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC)) ensure_legacy(c);  // not inside the graph capture
  if (int rc = ck_autotune(c)) return rc;
  if (int rc = ckh_autotune(c)) return rc;
  if (ck_active(c) || ckh_active(c)) ensure_ck_w(c);
  if (int rc = res_autotune(c, m, q_tol, r_tol)) return rc;
  const bool norms = q_tol > 0 || r_tol > 0;
  const bool p2p_terms = c->p2p && !c->joint && c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC;
  const bool use_res = m > 0 && m <= 250 && res_active(c);  // (a granule tag carries the term in 8 bits)
  if (int rc = run_series(c, m, q_tol, r_tol, use_res)) return rc;
  int iters = m, status = POVAR_LINEAR_SOLVER_NO_CONVERGENCE;
  if (p2p_terms) c->flag0_clean = false;  // the waits of the exchange kernels raise bit 1 of flags[0] on a time-out
  if (c->deterministic && (ck_active(c) || ckh_active(c))) {  // e0_ck[_h]_det: a ticket that never came up raises bit 3 (bounded spins):
    c->flag0_clean = false;                // looked at where the caller next waits for the series (res_verify)
    c->det_check = true;
  }
  if (p2p_terms) {
    int f[4];
    if (int rc = read_flags(c, f)) return rc;
    if (f[0] & 2) {
      HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(int), c->stream));
      return fail(-3, "peer-to-peer exchange: a rank did not deliver its partial sums (wait timed out)");
    }
  }
  if (norms) {
    if (int rc = res_verify(c)) return rc;  // (the read-back below must see the flags of a complete series)
    int f[4];
    if (int rc = read_flags(c, f)) return rc;
    if (f[1]) {
      iters = f[2];
      status = POVAR_LINEAR_SOLVER_SUCCESS;
    }
  }
  if (num_iterations) *num_iterations = iters;
  if (termination) *termination = status;
  return 0;
}

int povar_get_increment(povar_ctx* c, double* inc) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  return read_cam_vector(c, inc, c->accum.p, (size_t)(c->joint ? 11 : 12) * c->n_cams);
}

int povar_get_term(povar_ctx* c, double* term) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  return read_cam_vector(c, term, c->tmp.p, (size_t)(c->joint ? 11 : 12) * c->n_cams);
}

int povar_solve_pose(povar_ctx* c, double lambda, int32_t solver_type, int32_t m, double q_tol,
                     double r_tol, double* inc, int32_t* num_iterations, int32_t* termination) {
  if (int rc = povar_prepare_pose(c, lambda, solver_type)) return rc;
  if (int rc = povar_power_series_pose(c, m, q_tol, r_tol, num_iterations, termination)) return rc;
  if (int rc = povar_get_increment(c, inc)) return rc;
  for (size_t i = 0; i < 12 * (size_t)c->n_cams; ++i)
    if (!std::isfinite(inc[i])) return POVAR_NUMERIC_FAILURE;  // bal_bundle_adjustment.cpp:362
  return 0;
}

int povar_right_mul_e0_pose(povar_ctx* c, const double* x, double* y) {
  if (int rc = check_ctx(c)) return rc;
  const size_t n = 12 * (size_t)c->n_cams;
  HIP_TRY(hipMemcpyAsync(c->tmp.p, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));
  // z = sigma * x
  HIP_TRY(hipMemcpyAsync(c->inc.p, c->tmp.p, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(cam_apply_inc, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d, 1);
  int mode = 1;
  if (int rc = launch_e0(c, &mode)) return rc;
  if (mode == 1 || mode == 3) {
    const Dp dt = mode == 3 ? ldsacc_dp(c) : c->d;
    hipLaunchKernelGGL(cam_sum_items, dim3(grid_for(c->n_cams, 4)), dim3(256), 0, c->stream, dt, c->d.y, 1);
  }
  HIP_TRY(hipMemcpyAsync(y, c->y.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemsetAsync(c->y.p, 0, sizeof(double) * n, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

int povar_solve_joint(povar_ctx* c, double lambda, int32_t m, double q_tol, double r_tol, double* inc,
                      int32_t* num_iterations, int32_t* termination) {
  if (int rc = povar_prepare_joint(c, lambda)) return rc;
  if (int rc = povar_power_series_pose(c, m, q_tol, r_tol, num_iterations, termination)) return rc;
  if (int rc = povar_get_increment(c, inc)) return rc;
  for (size_t i = 0; i < 11 * (size_t)c->n_cams; ++i)
    if (!std::isfinite(inc[i])) return POVAR_NUMERIC_FAILURE;
  return 0;
}

int povar_set_e0_mode(povar_ctx* c, int32_t mode) {
  if (int rc = check_ctx(c)) return rc;
  if (mode != POVAR_E0_IMPLICIT && mode != POVAR_E0_TILES && mode != POVAR_E0_IMPLICIT_LDSACC &&
      mode != POVAR_E0_TILES_LDSACC) return fail(-1, "bad e0 mode");
  if (c->deterministic) return 0;  // pinned (POVAR_DETERMINISTIC)
  c->opt.e0_mode = mode;
  if (c->linearized) return ensure_tiles(c);
  return 0;
}

int povar_profile_enable(povar_ctx* c, int32_t enable) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->profile = enable != 0;
  c->ev_used = 0;
  return 0;
}

int povar_profile_get(povar_ctx* c, povar_profile_info* out) {
  if (int rc = check_ctx(c)) return rc;
  if (!out) return fail(-1, "null argument");
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::memset(out, 0, sizeof(*out));
  for (size_t i = 0; i + 1 < c->ev_used; ++i) {
    const int kind = c->ev_kind[i];
    if (kind < 0) continue;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
    if (kind == 0) { out->e0_ms += ms; out->e0_launches++; }
    else if (kind == 1) { out->binv_ms += ms; out->binv_launches++; }
    else { out->comm_ms += ms; out->comm_launches++; }
  }
  c->ev_used = 0;
  return 0;
}

int povar_e0_model_bytes(povar_ctx* c, int64_t* lm_kernel, int64_t* cam_kernel) {
  if (int rc = check_ctx(c)) return rc;
  if (!lm_kernel || !cam_kernel) return fail(-1, "null argument");
  // Bytes the E0 kernels of the current mode must move per application BY DESIGN (every array they stream, once;
  // arrays that stay in L2 -- camera records, z, the LDS image -- counted once, not per gather).  This is the
  // byte floor bench.py prices the measured kernel time against; the PMC-measured traffic is reported beside it.
  const int64_t ns = c->n_slots, nl = c->n_lms, nc = c->n_cams, no = c->n_obs;
  const int64_t robust = c->opt.robust_norm ? 8 : 0;
  const int64_t cam_static = nc * (96 + 96);            // z (12 doubles) + P (12 doubles) per camera
  const bool acc = c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC || c->opt.e0_mode == POVAR_E0_TILES_LDSACC;
  const bool lik = c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC && c->long_in_kernel;
  const bool lpl = c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC;
  const int64_t n_cold = lpl ? c->n_cold3 : lik ? c->n_cold2 : c->n_cold;
  const int64_t hot_flush = lpl ? (int64_t)c->v2_part.n * 8 : acc ? (int64_t)c->e0c_grid * c->n_hot_acc * 96 : 0;
  const int64_t tail = nc * (1152 + 96 /*sigma*/ + 3 * 96 /*accum rw, tmp*/ + 96 /*z*/);
  int64_t lm = 0, cm = 0;
  if (c->deterministic && ck_active(c)) {
    // e0_ck_det: as e0_ck below with the rows THREE times (two walks forward, one back), the landmark records twice (h~, G),
    // and 1 + 2 bytes of counts / tickets per landmark lane / chunk lane
    const int64_t part = (int64_t)c->ck.n_part_rec * 96;
    *lm_kernel = 3 * c->ck.rows * WAVE * 18 + (int64_t)c->d.v2.n_tiles * WAVE * (72 + 1) + cam_static +
                 4 * (int64_t)(c->ck.lane_meta.n) * 8 + (int64_t)c->ck.lane_meta.n * 2 + part;
    *cam_kernel = part + tail;
    return 0;
  }
  if (c->deterministic && ckh_active(c)) {  // e0_ck_h_det: the 2-byte rows (+ 8 with a robust norm) three times, X / records twice
    const int64_t part = (int64_t)c->ckh.n_part_rec * 96;
    *lm_kernel = 3 * c->ckh.rows * WAVE * (2 + robust) + (int64_t)c->d.v2.n_tiles * WAVE * (112 + 1) + cam_static +
                 4 * (int64_t)(c->ckh.lane_meta.n) * 8 + (int64_t)c->ckh.lane_meta.n * 2 + part;
    *cam_kernel = part + tail;
    return 0;
  }
  switch (c->opt.e0_mode) {
    case POVAR_E0_IMPLICIT_LDSACC:
      if (ck_active(c)) {
        // e0_ck: the chunk rows (uv 16 + landmark slot 2 bytes) on BOTH passes -- the kernel as built reads them twice --,
        // the 72-byte landmark records once, 8 bytes of lane metadata per chunk lane and pass, the partial records out
        // (one per workgroup slot + one per chunk of a camera without a slot); the per-camera kernel reads those back
        const int64_t part = (int64_t)c->ck.n_part_rec * 96;
        lm = 2 * c->ck.rows * WAVE * (c->ck.packed ? 10 : 18) /* image point 16 bytes (8 packed) + slot 2; no weight array: recomputed (ck_huber_w) */ + (int64_t)c->d.v2.n_tiles * WAVE * 72 + cam_static +
             2 * (int64_t)(c->ck.lane_meta.n) * 8 + part;
        cm = part + tail;
        if (c->ck.cold_q) {  // cold observations: position 4 + q 32 bytes out of e0_ck, q 32 + landmark copy 24 into the per-camera kernel
          lm += 36 * c->n_cold3;
          cm += 56 * c->n_cold3;
        }
        break;
      }
      if (ckh_active(c)) {
        // e0_ck_h: landmark slot (2 bytes) [+ weight] per observation on both passes -- the step-2 operator does not read
        // the image coordinates --, the 112-byte landmark records once, lane metadata, partial records out and back
        const int64_t part = (int64_t)c->ckh.n_part_rec * 96;
        lm = 2 * c->ckh.rows * WAVE * (2 + robust) + (int64_t)c->d.v2.n_tiles * WAVE * 112 + cam_static +
             2 * (int64_t)(c->ckh.lane_meta.n) * 8 + part;
        cm = part + tail;
        break;
      }
      if (c->use_lpl)  // e0_lpl: uv + camera slot per row slot, 72-byte landmark records; e0_lpl_h: the camera slot only (its
                       // operator does not depend on uv: the loads are dead code), 112-byte records; cold: position + q out
        lm = c->v2_rows * WAVE * ((c->joint ? 4 : 20) + robust) + (int64_t)c->d.v2.n_tiles * WAVE * (c->joint ? 112 : 72) + cam_static +
             n_cold * (c->q_rows ? 32 : 36) + hot_flush;  // q_rows: no position load, the per-camera kernel reads the index
      else
          lm = ns * (E0_SLOT_BYTES + robust) + nl * E0_LMREC_BYTES + cam_static + n_cold * 32 + hot_flush;
      cm = hot_flush + n_cold * (32 + 24 + (lpl && c->q_rows ? 4 : 0)) + tail;
      break;
    case POVAR_E0_IMPLICIT:
      lm = ns * (28 + robust) + nl * 96 + cam_static + no * 32;   // uv, cam, lm, meta; q4 out
      cm = no * (4 + 32 + 24) + tail;                                // cm_slot, q4 gather, cm_h
      break;
    case POVAR_E0_TILES:
      lm = ns * (12 + 480 + robust) + nl * 72 + nc * 96 + no * 32;
      cm = no * (4 + 32 + 24) + tail;
      break;
    case POVAR_E0_TILES_LDSACC:
      lm = ns * (12 + 480 + robust) + nl * 72 + nc * 96 + n_cold * 32 + hot_flush;
      cm = hot_flush + n_cold * (4 + 32 + 24) + tail;
      break;
    default:
      return fail(-1, "bad e0 mode");
  }
  *lm_kernel = lm;
  *cam_kernel = cm;
  return 0;
}

int povar_set_e0_kernel(povar_ctx* c, int32_t kernel) {
  if (int rc = check_ctx(c)) return rc;
  if (kernel < -1 || kernel > CK_VARIANTS) return fail(-1, "unknown E0 kernel");
  if (kernel > 0 && !c->ck_zero_range.p) return fail(-1, "the camera-chunk layout was not built for this context");
  if (c->deterministic) return 0;  // pinned (POVAR_DETERMINISTIC)
  if (kernel < 0) {  // back to the library's own choice
    c->ck_auto = true;
    c->ck_tuned = c->ckh_tuned = false;
    c->ck_variant = c->ckh_variant = 0;
    return 0;
  }
  c->ck_auto = false;
  c->ck_variant = kernel;
  c->ckh_variant = kernel > 0 ? 1 : 0;  // (step 2 has one camera-chunk instantiation)
  return 0;
}

int povar_set_series_kernel(povar_ctx* c, int32_t mode) {
  if (int rc = check_ctx(c)) return rc;
  if (mode < -1 || mode > 1) return fail(-1, "unknown series kernel");
  if (mode == 1 && !c->res.ready) return fail(-1, "the resident-series layout was not built for this context");
  if (int rc = res_verify(c)) return rc;
  if (c->deterministic) return 0;  // pinned (POVAR_DETERMINISTIC)
  c->res_mode = mode;
  if (mode < 0) c->res_tuned = false;
  return 0;
}

int povar_debug_ck_stamps(povar_ctx* c, uint64_t* out, int64_t n) {
  // (the diagnostic build -- tools/variants/ck_stamps.patch -- replaces this body; the shipped kernels execute no stamp)
  (void)c; (void)out; (void)n;
  return fail(-1, "diagnostic builds only (tools/variants/build_variant.sh ck_stamps)");
}

}  // extern "C"
