// povar_lm.hip -- the stages of an LM iteration behind the C ABI: state, cost, linearise, prepare, apply (step 1 and step 2), exports.
#include "povar_ctx.hpp"

// every kernel of the LM iteration runs on the lane-per-landmark layout: nothing reads the legacy camera-major copies
bool lpl_only(const povar_ctx* c) {
  return c->use_lpl && c->use_lpl_prepare && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC;
}

void build_views(povar_ctx* c) {
  const int hom = c->linearized_h ? 1 : 0;
  hipLaunchKernelGGL(cm_build_h, dim3(grid_for(c->n_obs, 256)), dim3(256), 0, c->stream, c->d, (const int*)c->cm_lm.p, c->cm_h.p, c->n_obs, hom);
  if (c->n_cold > 0)
    hipLaunchKernelGGL(cm_build_h, dim3(grid_for(c->n_cold, 256)), dim3(256), 0, c->stream, c->d, (const int*)c->cc_lm.p, c->cc_h.p, c->n_cold, hom);
  if (c->long_in_kernel && c->n_cold2 > 0)
    hipLaunchKernelGGL(cm_build_h, dim3(grid_for(c->n_cold2, 256)), dim3(256), 0, c->stream, c->d, (const int*)c->c2_lm.p, c->c2_h.p, c->n_cold2, hom);
  c->views_lin_id = c->lin_id;
}

// called by every entry point that may run a lane-per-observation ("legacy") kernel: the camera-major landmark copies
// (cm_scatter, the cold views of e0_lm_cached) and the per-slot sqrt(w) / weighted residual arrays, which the
// lane-per-landmark linearisation (lpl_pass<0>) does not write
// lane-ordered mirrors (V2::lmx / lml / lsc), rebuilt from the landmark-order masters when stale
void lanes_from(povar_ctx* c, const double4* src, double4* dst) {
  const int64_t n = (int64_t)c->d.v2.n_tiles * WAVE;
  if (n > 0) hipLaunchKernelGGL(lm_to_lanes, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d.v2.lm_of, src, dst, n);
}

void ensure_lmx(povar_ctx* c) {
  if (c->lmx_ver == c->lms_ver) return;
  lanes_from(c, c->lms4.p, c->v2_lmx.p);
  c->lmx_ver = c->lms_ver;
}

void ensure_lin_mirrors(povar_ctx* c) {
  if (c->lml_lin_id != c->lin_id) { lanes_from(c, c->lms_lin4.p, c->v2_lml.p); c->lml_lin_id = c->lin_id; }
  if (c->lsc_lin_id != c->lin_id) { lanes_from(c, c->jl_scale4.p, c->v2_lsc.p); c->lsc_lin_id = c->lin_id; }
}

void ensure_jl_scale4(povar_ctx* c) {
  if (c->jls_lin_id == c->lin_id) return;
  const int64_t n = (int64_t)c->d.v2.n_tiles * WAVE;
  if (n > 0) hipLaunchKernelGGL(lanes_to_lm, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d.v2.lm_of, c->d.v2.seg,
                                (const double4*)c->v2_lsc.p, c->jl_scale4.p, n);
  c->jls_lin_id = c->lin_id;
}

// the landmark-order copy of the linearisation point: the lane-per-landmark linearisation keeps only the lane-ordered
// one (V2::lml); the lane-per-observation kernels and the exports read lms_lin4
void ensure_lms_lin(povar_ctx* c) {
  if (c->lmslin_lin_id == c->lin_id) return;
  const int64_t n = (int64_t)c->d.v2.n_tiles * WAVE;
  if (n > 0) hipLaunchKernelGGL(lanes_to_lm, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d.v2.lm_of, c->d.v2.seg,
                                (const double4*)c->v2_lml.p, c->lms_lin4.p, n);
  c->lmslin_lin_id = c->lin_id;
}

void ensure_legacy(povar_ctx* c) {
  if (!(c->linearized || c->linearized_h)) return;
  ensure_jl_scale4(c);
  ensure_lms_lin(c);
  c->flag0_clean = false;  // the auxiliary linearisation below may raise the finiteness flag
  if (c->views_lin_id != c->lin_id) build_views(c);
  // the lazily rebuilt sqrt(w) / residual arrays and landmark records belong to the LINEARISATION: they are built
  // with its alpha, whatever alpha the caller (apply_pose, error_pose) has put into the context meanwhile
  struct AlphaGuard {
    povar_ctx* c;
    double sa, sb;
    explicit AlphaGuard(povar_ctx* c_) : c(c_), sa(c_->d.sa), sb(c_->d.sb) {
      if (c->linearized && !c->linearized_h) {
        c->d.sa = std::sqrt(c->alpha_lin);
        c->d.sb = std::sqrt(1.0 - c->alpha_lin);
      }
    }
    ~AlphaGuard() { c->d.sa = sa; c->d.sb = sb; }
  } guard(c);
  if (c->aux_lin_id != c->lin_id) {
    Dp da = c->d;
    da.lin_aux_only = 1;
    if (c->linearized_h) {
      hipLaunchKernelGGL((lm_regular<OpLinearizeH>), dim3(c->n_reg_blocks), dim3(LM_BLOCK), 0, c->stream, da, OpLinearizeH{}, c->part.p);
      if (c->n_long > 0)
        hipLaunchKernelGGL((lm_long<OpLinearizeH>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, da, OpLinearizeH{}, c->part.p);
    } else {
      hipLaunchKernelGGL((lm_regular<OpLinearize>), dim3(c->n_reg_blocks), dim3(LM_BLOCK), 0, c->stream, da, OpLinearize{}, c->part.p);
      if (c->n_long > 0)
        hipLaunchKernelGGL((lm_long<OpLinearize>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, da, OpLinearize{}, c->part.p);
    }
    c->aux_lin_id = c->lin_id;
  }
  if (c->prep_id && c->aux_prep_id != c->prep_id && c->prep_lin_id == c->lin_id) {
    // the landmark half of prepare_Hb again, on the lane-per-observation layout: Hll^-1 and the packed landmark records
    Dp da = c->d;
    da.prep_aux_only = 1;
    if (c->joint) {
      hipLaunchKernelGGL((lm_regular<OpPrepareH>), dim3(c->n_reg_blocks), dim3(LM_BLOCK), 0, c->stream, da, OpPrepareH{}, c->part.p);
      if (c->n_long > 0)
        hipLaunchKernelGGL((lm_long<OpPrepareH>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, da, OpPrepareH{}, c->part.p);
    } else {
      hipLaunchKernelGGL((lm_regular<OpPrepare>), dim3(c->n_reg_blocks), dim3(LM_BLOCK), 0, c->stream, da, OpPrepare{}, c->part.p);
      if (c->n_long > 0)
        hipLaunchKernelGGL((lm_long<OpPrepare>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, da, OpPrepare{}, c->part.p);
    }
    c->aux_prep_id = c->prep_id;
  }
}

// flags[0] (finiteness / p2p time-out bits) is reset before every entry point that reads it back; when the last
// read-back was zero and nothing that can raise it has been enqueued since, the reset is skipped
int clear_flag0(povar_ctx* c) {
  if (!c->flag0_clean) HIP_TRY(hipMemsetAsync(c->flags.p, 0, sizeof(int), c->stream));
  c->flag0_clean = false;  // writers follow; clean again only after a read-back of zero (not when that read-back fails: ADVICE r03)
  return 0;
}

bool err_memo_hit(const povar_ctx* c, int kind, double alpha, povar_residual_info* out) {
  // (not while sharded: a hit returns before the all-reduce of the failure flag, and whether a rank's memo is valid
  // depends on rank-local events -- a rank that recomputes would enter the collective alone.  ADVICE r03.)
  if (sharded(c)) return false;
  const auto& m = c->err_memo;
  if (!m.valid || m.kind != kind || m.alpha != alpha || m.lms_ver != c->lms_ver || m.cams_ver != c->cams_ver ||
      m.mode != c->opt.e0_mode * 4 + (c->use_lpl ? 2 : 0) + (c->use_lpl_prepare ? 1 : 0))
    return false;
  *out = m.ri;
  return true;
}

void err_memo_store(povar_ctx* c, int kind, double alpha, const povar_residual_info& ri) {
  auto& m = c->err_memo;
  m.valid = !c->no_err_memo;
  m.kind = kind;
  m.alpha = alpha;
  m.lms_ver = c->lms_ver;
  m.cams_ver = c->cams_ver;
  m.mode = c->opt.e0_mode * 4 + (c->use_lpl ? 2 : 0) + (c->use_lpl_prepare ? 1 : 0);
  m.ri = ri;
}

// OR of a per-rank failure flag over the ranks (is_numerically_valid, linearisation failure)
int combine_flag(povar_ctx* c, int* flag) {
  if (!sharded(c)) return 0;
  double v = *flag ? 1.0 : 0.0;
  HIP_TRY(hipMemcpyAsync(c->scal.p + 7, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (int rc = allreduce(c, c->scal.p + 7, 1)) return rc;
  HIP_TRY(hipMemcpyAsync(&v, c->scal.p + 7, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *flag = v > 0 ? 1 : 0;
  return 0;
}

int ensure_tiles(povar_ctx* c) {
  if (c->opt.e0_mode != POVAR_E0_TILES && c->opt.e0_mode != POVAR_E0_TILES_LDSACC) return 0;
  if (!c->tiles.p) {
    HIP_TRY(c->tiles.alloc((size_t)c->n_bins * TILE_PAIRS * WAVE, &c->bytes));
    c->d.tiles = c->tiles.p;
    c->tiles_valid = false;
  }
  if (!c->tiles_valid) {
    ensure_legacy(c);
    hipLaunchKernelGGL(materialize_tiles, dim3(grid_for(c->n_slots, LM_BLOCK)), dim3(LM_BLOCK), 0,
                       c->stream, c->d);
    c->tiles_valid = true;
  }
  return 0;
}

extern "C" {

int povar_set_cameras(povar_ctx* c, const double* cams) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->cams4.p, cams, sizeof(double) * 12 * c->n_cams, hipMemcpyHostToDevice, c->stream));
  ++c->cams_ver;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int povar_get_cameras(povar_ctx* c, double* cams) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipMemcpyAsync(cams, c->cams4.p, sizeof(double) * 12 * c->n_cams, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int povar_set_landmarks(povar_ctx* c, const double* lms) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->stage.p, lms, sizeof(double) * 3 * c->n_lms, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(lms3_to_4, dim3(grid_for(c->n_lms, 256)), dim3(256), 0, c->stream, c->stage.p,
                     c->lms4.p, c->n_lms);
  ++c->lms_ver;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int povar_get_landmarks(povar_ctx* c, double* lms) {
  if (int rc = check_ctx(c)) return rc;
  hipLaunchKernelGGL(lms4_to_3, dim3(grid_for(c->n_lms, 256)), dim3(256), 0, c->stream, c->lms4.p,
                     c->stage.p, c->n_lms);
  HIP_TRY(hipMemcpyAsync(lms, c->stage.p, sizeof(double) * 3 * c->n_lms, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int povar_backup_pose(povar_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->cams_bak4.p, c->cams4.p, sizeof(double4) * 3 * c->n_cams, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->lms_bak4.p, c->lms4.p, sizeof(double4) * c->n_lms, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int povar_restore_pose(povar_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->cams4.p, c->cams_bak4.p, sizeof(double4) * 3 * c->n_cams, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->lms4.p, c->lms_bak4.p, sizeof(double4) * c->n_lms, hipMemcpyDeviceToDevice, c->stream));
  ++c->lms_ver;
  ++c->cams_ver;
  return 0;
}

void set_alpha(povar_ctx* c, double alpha) {
  c->d.sa = std::sqrt(alpha);
  c->d.sb = std::sqrt(1.0 - alpha);
}

int povar_init_landmarks_pose(povar_ctx* c, double alpha) {
  if (int rc = check_ctx(c)) return rc;
  ++c->lms_ver;
  set_alpha(c, alpha);
  TimeScope ts(c, 4);
  if (c->k1_qr) {
    hipLaunchKernelGGL(init_landmarks_qr, dim3(grid_for(c->n_lms, 256)), dim3(256), 0, c->stream, c->d);
  } else {
    launch_lm(c, OpInit{});
    launch_lm(c, OpInitRefine{});
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

int povar_error_pose(povar_ctx* c, double alpha, povar_residual_info* out) {
  if (int rc = check_ctx(c)) return rc;
  if (!out) return fail(-1, "null argument");
  TimeScope ts(c, 4);
  set_alpha(c, alpha);
  if (err_memo_hit(c, 1, alpha, out)) return 0;
  if (int rc = clear_flag0(c)) return rc;
  if (c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC) {
    ensure_lmx(c);
    hipLaunchKernelGGL(lpl_pass<1>, dim3(c->e0c_grid), dim3(E0C_BLOCK), pass_lds_bytes(c->v2_max_slots), c->stream, c->d, c->part.p);
    hipLaunchKernelGGL((reduce_partials<3>), dim3(1), dim3(1024), 0, c->stream, c->part.p, c->e0c_grid, c->scal.p);
  } else {
    launch_lm(c, OpError{});
    launch_reduce<3>(c, c->scal.p);
  }
  HIP_TRY(hipGetLastError());
  if (int rc = allreduce(c, c->scal.p, 3)) return rc;
  double h[3];
  int f[4];
  if (int rc = read_scal_flags(c, h, 3, f)) return rc;
  c->flag0_clean = f[0] == 0;
  if (int rc = combine_flag(c, &f[0])) return rc;
  out->all_num_obs = (int64_t)std::llround(h[2]);
  out->all_error = h[0];
  out->all_residual_sum = h[1];
  out->valid_num_obs = out->all_num_obs;  // projection_valid is always true on pOSE (helper.cpp:263)
  out->valid_error = h[0];
  out->valid_residual_sum = h[1];
  out->is_numerically_valid = f[0] ? 0 : 1;
  err_memo_store(c, 1, alpha, *out);
  return 0;
}

int povar_linearize_pose(povar_ctx* c, double alpha) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  c->linearized_h = false;
  set_alpha(c, alpha);
  c->alpha_lin = alpha;
  if (int rc = swap_in_placed_rows(c, false); rc < 0) return rc;  // a new linearisation point: the row order may change
  TimeScope ts(c, 0);
  if (int rc = clear_flag0(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->cams_lin4.p, c->cams4.p, sizeof(double4) * 3 * c->n_cams, hipMemcpyDeviceToDevice, c->stream));
  ++c->lin_id;
  c->linearized = true;
  // lane-per-landmark mode: one forward walk over the row stream; the per-slot arrays and the camera-major landmark
  // copies of the lane-per-observation kernels are built when one of them asks (ensure_legacy)
  const bool lazy = lpl_only(c);
  Dp dl = c->d;  // the camera-major kernels of this call read the linearisation point where it is now
  if (lazy) {
    // the kernel reads the current lane-ordered mirror and leaves the linearisation point (V2::lml) and the scale
    // mirror behind; the landmark-order copy lms_lin4 follows when a lane-per-observation kernel asks
    // (ensure_lms_lin): until then the current landmarks ARE the linearisation point
    ensure_lmx(c);
    c->lml_lin_id = c->lsc_lin_id = c->lin_id;
    hipLaunchKernelGGL(lpl_pass<0>, dim3(c->e0c_grid), dim3(E0C_BLOCK), pass_lds_bytes(c->v2_max_slots), c->stream, c->d, c->part.p);
    if (c->has_empty_lm) {
      HIP_TRY(hipMemcpyAsync(c->lms_lin4.p, c->lms4.p, sizeof(double4) * c->n_lms, hipMemcpyDeviceToDevice, c->stream));
      c->lmslin_lin_id = c->lin_id;
    } else {
      dl.lms_lin4 = c->lms4.p;
    }
  } else {
    HIP_TRY(hipMemcpyAsync(c->lms_lin4.p, c->lms4.p, sizeof(double4) * c->n_lms, hipMemcpyDeviceToDevice, c->stream));
    c->lmslin_lin_id = c->lin_id;
    launch_lm(c, OpLinearize{});
    c->aux_lin_id = c->jls_lin_id = c->lin_id;
    build_views(c);
  }
  if (c->n_cold3 > 0)
    hipLaunchKernelGGL(cm_build_h, dim3(grid_for(c->n_cold3, 256)), dim3(256), 0, c->stream, dl, (const int*)c->c3_lm.p, c->c3_h.p, c->n_cold3, 0);
  hipLaunchKernelGGL(cm_gram, dim3(grid_for(c->n_items, 4)), dim3(256), 0, c->stream, dl, lazy ? 1 : 0);
  if (sharded(c)) {
    // per-camera Gram moments are partial sums over this rank's landmarks: sum, all-reduce, finish
    hipLaunchKernelGGL(cam_finish_linearize, dim3(c->n_cams), dim3(CFL_THREADS), 0, c->stream, c->d, (const double*)nullptr);
    if (int rc = allreduce(c, c->d.G, 40 * (size_t)c->n_cams)) return rc;
    hipLaunchKernelGGL(cam_finish_linearize, dim3(c->n_cams), dim3(CFL_THREADS), 0, c->stream, c->d, (const double*)c->d.G);
  } else {
    hipLaunchKernelGGL(cam_finish_linearize, dim3(c->n_cams), dim3(CFL_THREADS), 0, c->stream, c->d, (const double*)nullptr);
  }
  HIP_TRY(hipGetLastError());
  int f[4];
  if (int rc = read_flags(c, f)) return rc;
  c->flag0_clean = f[0] == 0;
  if (int rc = combine_flag(c, &f[0])) return rc;
  c->new_linearization_point = true;
  c->linearized = true;
  c->tiles_valid = false;
  return f[0] ? POVAR_NUMERIC_FAILURE : 0;
}

int povar_prepare_pose(povar_ctx* c, double lambda, int32_t solver_type) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  if (!c->linearized) return fail(-1, "povar_prepare_pose before povar_linearize_pose");
  set_alpha(c, c->alpha_lin);
  c->joint = false;
  ++c->prep_id;
  c->prep_lin_id = c->lin_id;
  TimeScope ts(c, 1);
  // scale_Jp_cols_pOSE on a new linearisation point (linearizor_power_varproj.cpp:192-195):
  // the scaling is part of the implicit tile; only stored tiles need (re)materialising.
  c->new_linearization_point = false;
  c->d.lambda_lm = solver_type == POVAR_POWER_SCHUR_COMPLEMENT ? lambda : 0.0;  // cpp:197-200
  hipLaunchKernelGGL(build_hot_rec, dim3(grid_for((int64_t)c->n_cams * 12, 256)), dim3(256), 0, c->stream, c->d, 0);
  if (c->use_lpl && c->use_lpl_prepare && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC) {
    // lane-per-landmark K7: Hll^-1, landmark records and the per-camera partial sums of b in one kernel, then the
    // per-camera sum of the partials and the cold observations (same kernel as the per-term one, output b)
    Dp da = ldsacc_dp(c, true);
    da.prep_lpl_only = 1;
    ensure_lin_mirrors(c);
    // cam_cold_sum honours the series-done flag of the term loop: clear what an early exit of the last solve left
    HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));
    if (c->opt.robust_norm)
      hipLaunchKernelGGL(prepare_lpl<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK), prep_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    else
      hipLaunchKernelGGL(prepare_lpl<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK), prep_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    da.y = c->d.b;
    da.p2p_peer = nullptr;  // b goes through the ordinary exchange below, not the per-term push
    da.p2p_epoch = nullptr;
    hipLaunchKernelGGL(cam_cold_sum<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, da, 0);
  } else {
    c->aux_prep_id = c->prep_id;  // this branch writes them
    ensure_legacy(c);
    launch_lm(c, OpPrepare{});
    hipLaunchKernelGGL(cm_scatter, dim3(grid_for(c->n_items, 4)), dim3(256), 0, c->stream, c->d, 0, 0);
    hipLaunchKernelGGL(cam_sum_items, dim3(grid_for(c->n_cams, 4)), dim3(256), 0, c->stream, c->d, c->d.b, 1);
  }
  if (int rc = allreduce(c, c->d.b, 12 * (size_t)c->n_cams)) return rc;
  hipLaunchKernelGGL(cam_build_binv, dim3(grid_for(c->n_cams, K8_CAMS_PER_WG)), dim3(K8_THREADS), 0, c->stream,
                     c->d, lambda);
  if (int rc = ensure_tiles(c)) return rc;
  // the one-off choice between the step-1 term kernels is part of the preparation, not of the first solve's time
  // (solve_reduced_system_time of the caller's log: bal_bundle_adjustment.cpp:355-360)
  if (int rc = ck_autotune(c)) return rc;
  if (int rc = tune_agree(c, 0)) return rc;
  HIP_TRY(hipGetLastError());
  return 0;
}

int povar_apply_pose(povar_ctx* c, int32_t solver_type, double alpha, const double* inc, double* l_diff) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  if (!c->linearized) return fail(-1, "povar_apply_pose before povar_linearize_pose");
  const size_t n = 12 * (size_t)c->n_cams;
  set_alpha(c, alpha);
  TimeScope ts(c, 3);
  if (int rc = write_cam_vector(c, c->inc.p, inc, n)) return rc;
  bool lpl_back = false;
  if (solver_type == POVAR_POWER_VARPROJ) {
    // cpp:250-256: scale, update cameras, unscale, back-substitute at the new cameras
    hipLaunchKernelGGL(cam_apply_inc, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d, 0);
    lpl_back = c->use_lpl && c->use_lpl_prepare && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC;
    if (lpl_back) {
      const Dp da = ldsacc_dp(c, true);
      ensure_lmx(c);
      ensure_lin_mirrors(c);
      if (c->opt.robust_norm)
        hipLaunchKernelGGL(backsub_lpl<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK), back_lds_bytes(c->v2_max_slots), c->stream, da, c->part.p);
      else
        hipLaunchKernelGGL(backsub_lpl<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK), back_lds_bytes(c->v2_max_slots), c->stream, da, c->part.p);
    } else {
      ensure_legacy(c);
      launch_lm(c, OpBackVarproj{});
    }
  } else {
    // cpp:260-270: back-substitute with the stored tiles, then update cameras
    hipLaunchKernelGGL(cam_apply_inc, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d, 1);
    ensure_legacy(c);
    launch_lm(c, OpBackPoba{});
    hipLaunchKernelGGL(cam_apply_inc, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->d, 2);
  }
  ++c->lms_ver;
  ++c->cams_ver;
  if (lpl_back) c->lmx_ver = c->lms_ver;  // backsub_lpl wrote the new landmarks into the lane-ordered mirror too
  if (lpl_back)
    hipLaunchKernelGGL((reduce_partials<1>), dim3(1), dim3(1024), 0, c->stream, c->part.p, c->e0c_grid, c->scal.p);
  else
    launch_reduce<1>(c, c->scal.p);
  HIP_TRY(hipGetLastError());
  if (int rc = allreduce(c, c->scal.p, 1)) return rc;
  double h = 0;
  if (int rc = read_scal(c, &h, 1)) return rc;
  if (l_diff) *l_diff = h;
  return 0;
}

// ------------------------------------------------------------------------------------------
// step 2
// ------------------------------------------------------------------------------------------
int povar_set_landmarks_homogeneous(povar_ctx* c, const double* lms_h) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->lms4.p, lms_h, sizeof(double) * 4 * c->n_lms, hipMemcpyHostToDevice, c->stream));
  ++c->lms_ver;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int povar_get_landmarks_homogeneous(povar_ctx* c, double* lms_h) {
  if (int rc = check_ctx(c)) return rc;
  HIP_TRY(hipMemcpyAsync(lms_h, c->lms4.p, sizeof(double) * 4 * c->n_lms, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int povar_backup_joint(povar_ctx* c) { return povar_backup_pose(c); }

int povar_restore_joint(povar_ctx* c) { return povar_restore_pose(c); }

int povar_error_homogeneous(povar_ctx* c, povar_residual_info* out) {
  if (int rc = check_ctx(c)) return rc;
  if (!out) return fail(-1, "null argument");
  TimeScope ts(c, 4);
  if (err_memo_hit(c, 2, 0.0, out)) return 0;
  if (int rc = clear_flag0(c)) return rc;
  if (c->use_lpl && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC) {
    ensure_lmx(c);
    hipLaunchKernelGGL(lpl_pass_h<1>, dim3(c->e0c_grid), dim3(E0C_BLOCK), pass_lds_bytes(c->v2_max_slots), c->stream, c->d, c->part.p);
    hipLaunchKernelGGL((reduce_partials<6>), dim3(1), dim3(1024), 0, c->stream, c->part.p, c->e0c_grid, c->scal.p);
  } else {
    launch_lm(c, OpErrorH{});
    launch_reduce<6>(c, c->scal.p);
  }
  HIP_TRY(hipGetLastError());
  if (int rc = allreduce(c, c->scal.p, 6)) return rc;
  double h[6];
  int f[4];
  if (int rc = read_scal_flags(c, h, 6, f)) return rc;
  c->flag0_clean = f[0] == 0;
  if (int rc = combine_flag(c, &f[0])) return rc;
  out->all_error = h[0];
  out->all_residual_sum = h[1];
  out->all_num_obs = (int64_t)std::llround(h[2]);
  out->valid_error = h[3];
  out->valid_residual_sum = h[4];
  out->valid_num_obs = (int64_t)std::llround(h[5]);
  out->is_numerically_valid = f[0] ? 0 : 1;
  err_memo_store(c, 2, 0.0, *out);
  return 0;
}

int povar_linearize_homogeneous(povar_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  if (int rc = swap_in_placed_rows(c, false); rc < 0) return rc;
  TimeScope ts(c, 0);
  if (int rc = clear_flag0(c)) return rc;
  HIP_TRY(hipMemcpyAsync(c->cams_lin4.p, c->cams4.p, sizeof(double4) * 3 * c->n_cams, hipMemcpyDeviceToDevice, c->stream));
  ++c->lin_id;
  c->linearized_h = true;
  const bool lazy = lpl_only(c);
  Dp dl = c->d;
  if (lazy) {
    // as in povar_linearize_pose: the kernel leaves the lane-ordered linearisation point and scale mirror behind,
    // jl_scale4 and lms_lin4 follow on demand
    ensure_lmx(c);
    c->lml_lin_id = c->lsc_lin_id = c->lin_id;
    hipLaunchKernelGGL(lpl_pass_h<0>, dim3(c->e0c_grid), dim3(E0C_BLOCK), pass_lds_bytes(c->v2_max_slots), c->stream, c->d, c->part.p);
    if (c->has_empty_lm) {
      HIP_TRY(hipMemcpyAsync(c->lms_lin4.p, c->lms4.p, sizeof(double4) * c->n_lms, hipMemcpyDeviceToDevice, c->stream));
      c->lmslin_lin_id = c->lin_id;
    } else {
      dl.lms_lin4 = c->lms4.p;
    }
  } else {
    HIP_TRY(hipMemcpyAsync(c->lms_lin4.p, c->lms4.p, sizeof(double4) * c->n_lms, hipMemcpyDeviceToDevice, c->stream));
    c->lmslin_lin_id = c->lin_id;
    launch_lm(c, OpLinearizeH{});
    c->aux_lin_id = c->jls_lin_id = c->lin_id;
    build_views(c);
  }
  if (c->n_cold3 > 0)
    hipLaunchKernelGGL(cm_build_h, dim3(grid_for(c->n_cold3, 256)), dim3(256), 0, c->stream, dl, (const int*)c->c3_lm.p, c->c3_h.p, c->n_cold3, 1);
  hipLaunchKernelGGL(cm_gram_h, dim3(grid_for(c->n_items, 4)), dim3(256), 0, c->stream, dl, lazy ? 1 : 0);
  hipLaunchKernelGGL(cam_finish_linearize_h, dim3(c->n_cams), dim3(CFL_THREADS), 0, c->stream, c->d, (const double*)nullptr, c->ncw.p);
  if (sharded(c)) {
    if (int rc = allreduce(c, c->d.G, 40 * (size_t)c->n_cams)) return rc;
    hipLaunchKernelGGL(cam_finish_linearize_h, dim3(c->n_cams), dim3(CFL_THREADS), 0, c->stream, c->d, (const double*)c->d.G, c->ncw.p);
  }
  HIP_TRY(hipGetLastError());
  int f[4];
  if (int rc = read_flags(c, f)) return rc;
  c->flag0_clean = f[0] == 0;
  if (int rc = combine_flag(c, &f[0])) return rc;
  c->new_linearization_point = true;
  c->linearized = false;  // the step-1 linearisation is gone
  c->linearized_h = true;
  c->tiles_valid = false;
  return f[0] ? POVAR_NUMERIC_FAILURE : 0;
}

int povar_prepare_joint(povar_ctx* c, double lambda) {
  if (int rc = check_ctx(c)) return rc;
  if (int rc = res_verify(c)) return rc;
  if (!c->linearized_h) return fail(-1, "povar_prepare_joint before povar_linearize_homogeneous");
  TimeScope ts(c, 1);
  c->joint = true;
  ++c->prep_id;
  c->prep_lin_id = c->lin_id;
  c->new_linearization_point = false;
  c->d.lambda_lm = lambda;  // set_landmark_damping_joint, linearizor_power_varproj.cpp:136
  hipLaunchKernelGGL(build_hot_rec, dim3(grid_for((int64_t)c->n_cams * 12, 256)), dim3(256), 0, c->stream, c->d, 1);
  if (c->use_lpl && c->use_lpl_prepare && c->opt.e0_mode == POVAR_E0_IMPLICIT_LDSACC) {
    // lane-per-landmark K7' (see povar_prepare_pose): landmark half + per-camera partials, per-camera sum of the
    // partials and the cold observations into the ambient 12-vector, then the tangent projection N_c^T
    Dp da = ldsacc_dp(c, true);
    da.prep_lpl_only = 1;
    ensure_lin_mirrors(c);
    HIP_TRY(hipMemsetAsync(c->flags.p + 1, 0, sizeof(int) * 3, c->stream));
    if (c->opt.robust_norm)
      hipLaunchKernelGGL(prepare_lpl_h<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK), prep_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    else
      hipLaunchKernelGGL(prepare_lpl_h<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK), prep_lds_bytes(c->v2_max_slots), c->stream, da, c->v2_part.p);
    da.y = c->d.y;
    da.p2p_peer = nullptr;
    da.p2p_epoch = nullptr;
    hipLaunchKernelGGL(cam_cold_sum<CCS_THREADS>, dim3(c->n_cams), dim3(CCS_THREADS), 0, c->stream, da, 1);
    hipLaunchKernelGGL(cam_nt_project, dim3(grid_for(c->n_cams, 256)), dim3(256), 0, c->stream, c->d, c->d.y, c->d.b,
                       (const double*)c->ncw.p);
  } else {
    c->aux_prep_id = c->prep_id;
    ensure_legacy(c);
    launch_lm(c, OpPrepareH{});
    hipLaunchKernelGGL(cm_scatter, dim3(grid_for(c->n_items, 4)), dim3(256), 0, c->stream, c->d, 0, 1);
    hipLaunchKernelGGL(cam_sum_items_h, dim3(grid_for(c->n_cams, 4)), dim3(256), 0, c->stream, c->d, c->d.b,
                       (const double*)c->ncw.p);
  }
  if (int rc = allreduce(c, c->d.b, 11 * (size_t)c->n_cams)) return rc;
  hipLaunchKernelGGL(cam_build_binv_h, dim3(grid_for(c->n_cams, K8_CAMS_PER_WG)), dim3(K8_THREADS), 0, c->stream, c->d,
                     lambda, (const double*)c->ncw.p);
  if (int rc = ckh_autotune(c)) return rc;  // (as in povar_prepare_pose: the one-off kernel choice is preparation)
  if (int rc = tune_agree(c, 1)) return rc;
  HIP_TRY(hipGetLastError());
  return 0;
}

int povar_apply_joint(povar_ctx* c, const double* inc, double* l_diff) {
  if (int rc = check_ctx(c)) return rc;
  if (!c->linearized_h) return fail(-1, "povar_apply_joint before povar_linearize_homogeneous");
  TimeScope ts(c, 3);
  HIP_TRY(hipMemcpyAsync(c->inc.p, inc, sizeof(double) * 11 * c->n_cams, hipMemcpyHostToDevice, c->stream));
  // cpp:280: back-substitute first (old cameras), then update the cameras (cpp:283-305)
  const bool lpl_back = lpl_only(c);
  if (lpl_back)  // the record image: P of the linearisation point (12..23), then z = sigma * N_c inc by cam_apply_inc_h (0..11)
    hipLaunchKernelGGL(build_hot_rec, dim3(grid_for((int64_t)c->n_cams * 12, 256)), dim3(256), 0, c->stream, c->d, 1);
  hipLaunchKernelGGL(cam_apply_inc_h, dim3(grid_for(c->n_cams, 256)), dim3(256), 0, c->stream, c->d, 1, (const double*)c->ncw.p);
  if (lpl_back) {
    const Dp da = ldsacc_dp(c, true);
    ensure_lmx(c);
    ensure_lin_mirrors(c);
    if (c->opt.robust_norm)
      hipLaunchKernelGGL(backsub_lpl_h<true>, dim3(c->e0c_grid), dim3(E0C_BLOCK), back_lds_bytes_h(c->v2_max_slots), c->stream, da, c->part.p);
    else
      hipLaunchKernelGGL(backsub_lpl_h<false>, dim3(c->e0c_grid), dim3(E0C_BLOCK), back_lds_bytes_h(c->v2_max_slots), c->stream, da, c->part.p);
  } else {
    ensure_legacy(c);
    launch_lm(c, OpBackJoint{});
  }
  ++c->lms_ver;
  ++c->cams_ver;
  if (lpl_back) c->lmx_ver = c->lms_ver;  // backsub_lpl_h keeps the lane-ordered mirror current
  hipLaunchKernelGGL(cam_apply_inc_h, dim3(grid_for(c->n_cams, 256)), dim3(256), 0, c->stream, c->d, 2, (const double*)c->ncw.p);
  if (lpl_back)
    hipLaunchKernelGGL((reduce_partials<1>), dim3(1), dim3(1024), 0, c->stream, c->part.p, c->e0c_grid, c->scal.p);
  else
    launch_reduce<1>(c, c->scal.p);
  HIP_TRY(hipGetLastError());
  if (int rc = allreduce(c, c->scal.p, 1)) return rc;
  double h = 0;
  if (int rc = read_scal(c, &h, 1)) return rc;
  if (l_diff) *l_diff = h;
  return 0;
}

int povar_normalize_joint(povar_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  // the lane-ordered mirror of the landmarks, when current, is normalised along (same division, same operands)
  const bool mirror = c->use_lpl && c->lmx_ver == c->lms_ver && c->d.v2.n_tiles > 0;
  const int64_t n_lanes = mirror ? (int64_t)c->d.v2.n_tiles * WAVE : 0;
  hipLaunchKernelGGL(normalize_joint, dim3(grid_for(std::max<int64_t>(std::max(c->n_cams, c->n_lms), n_lanes), 256)), dim3(256), 0,
                     c->stream, c->d, n_lanes);
  ++c->lms_ver;
  ++c->cams_ver;
  if (mirror) c->lmx_ver = c->lms_ver;
  HIP_TRY(hipGetLastError());
  return 0;
}

int povar_get_buffer(povar_ctx* c, int32_t which, double* out, int64_t n) {
  if (int rc = check_ctx(c)) return rc;
  ensure_legacy(c);  // exports rebuild the reference's tile from the per-slot arrays
  const size_t nc = c->n_cams, nl = c->n_lms;
  auto copy = [&](const void* src, size_t count) -> int {
    if ((size_t)n != count) return fail(-1, "povar_get_buffer: wrong size");
    HIP_TRY(hipMemcpyAsync(out, src, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
  };
  switch (which) {
    case POVAR_BUF_DIAG2: return copy(c->diag2.p, 12 * nc);
    case POVAR_BUF_POSE_SCALING: return copy(c->sigma.p, 12 * nc);
    case POVAR_BUF_HLL_INV: return copy(c->hll_inv.p, 9 * nl);
    case POVAR_BUF_B: return copy(c->b.p, 12 * nc);
    case POVAR_BUF_B_INV: return copy(c->binv.p, 144 * nc);
    case POVAR_BUF_B_JOINT: return copy(c->b.p, 11 * nc);
    case POVAR_BUF_B_INV_JOINT: {
      if ((size_t)n != 121 * nc) return fail(-1, "povar_get_buffer: wrong size");
      std::vector<double> h(144 * nc);
      HIP_TRY(hipMemcpyAsync(h.data(), c->binv.p, 144 * nc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      for (size_t k = 0; k < nc; ++k) std::memcpy(out + 121 * k, h.data() + 144 * k, 121 * sizeof(double));
      return 0;
    }
    case POVAR_BUF_NC_HOUSEHOLDER: return copy(c->ncw.p, 13 * nc);
    case POVAR_BUF_SC_PRECOND:
    case POVAR_BUF_SC_BLOCKDIAG: {
      const size_t dim2 = c->joint ? 121 : 144;
      if ((size_t)n != dim2 * nc) return fail(-1, "povar_get_buffer: wrong size");
      if (!c->sc_s.p) return fail(-1, "povar_get_buffer: no explicit-SC solve yet");
      std::vector<double> h(144 * nc);
      HIP_TRY(hipMemcpyAsync(h.data(), which == POVAR_BUF_SC_PRECOND ? c->sc_minv.p : c->sc_bmat.p,
                             144 * nc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      for (size_t k = 0; k < nc; ++k) std::memcpy(out + dim2 * k, h.data() + 144 * k, dim2 * sizeof(double));
      return 0;
    }
    case POVAR_BUF_JL_COL_SCALE_H: {
      if ((size_t)n != 4 * nl) return fail(-1, "povar_get_buffer: wrong size");
      HIP_TRY(hipMemcpyAsync(out, c->jl_scale4.p, nl * sizeof(double4), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      return 0;
    }
    case POVAR_BUF_JL_COL_SCALE: {
      if ((size_t)n != 3 * nl) return fail(-1, "povar_get_buffer: wrong size");
      std::vector<double4> h(nl);
      HIP_TRY(hipMemcpyAsync(h.data(), c->jl_scale4.p, nl * sizeof(double4), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      for (size_t l = 0; l < nl; ++l) {
        out[3 * l] = h[l].x;
        out[3 * l + 1] = h[l].y;
        out[3 * l + 2] = h[l].z;
      }
      return 0;
    }
    case POVAR_BUF_STORAGE: {
      if ((size_t)n != 64 * (size_t)c->n_obs) return fail(-1, "povar_get_buffer: wrong size");
      if (!c->linearized) return fail(-1, "not linearized");
      set_alpha(c, c->alpha_lin);
      const size_t cnt = (size_t)c->n_bins * TILE_PAIRS * WAVE;
      double2* tmp_tiles = nullptr;
      Dp d = c->d;
      if (!c->tiles.p) {
        HIP_TRY(hipMalloc((void**)&tmp_tiles, cnt * sizeof(double2)));
        d.tiles = tmp_tiles;
      }
      hipLaunchKernelGGL(materialize_tiles, dim3(grid_for(c->n_slots, LM_BLOCK)), dim3(LM_BLOCK), 0, c->stream, d);
      std::vector<double2> h(cnt);
      HIP_TRY(hipMemcpyAsync(h.data(), d.tiles, cnt * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      if (tmp_tiles) (void)hipFree(tmp_tiles);
      else c->tiles_valid = true;
      // blocked [bin][pair][lane] -> reference rows [4*obs + r][16] = [Jp(12) | Jl(3) | r]
      for (int64_t i = 0; i < c->n_obs; ++i) {
        const int s = c->slot_of_obs[i];
        const double2* t = h.data() + ((size_t)(s >> 6) * TILE_PAIRS) * WAVE + (s & 63);
        double v[64];
        for (int p = 0; p < TILE_PAIRS; ++p) {
          v[2 * p] = t[(size_t)p * WAVE].x;
          v[2 * p + 1] = t[(size_t)p * WAVE].y;
        }
        for (int r = 0; r < 4; ++r) {
          double* row = out + ((size_t)4 * i + r) * 16;
          for (int j = 0; j < 12; ++j) row[j] = v[12 * r + j];
          for (int j = 0; j < 3; ++j) row[12 + j] = v[48 + 3 * r + j];
          row[15] = v[60 + r];
        }
      }
      return 0;
    }
    default:
      return fail(-1, "povar_get_buffer: unknown buffer");
  }
}

}  // extern "C"
