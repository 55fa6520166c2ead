// povar_ctx.hpp -- what the translation units of the host side share: the context behind the opaque povar_ctx of
// include/povar_hip.h, error / buffer helpers, launch helpers that are templates, and the declarations of the functions one
// unit defines and another calls.  The host side of the C ABI is cut by concern:
//   povar_create.hip   layout construction and upload, povar_create / povar_destroy, read-back helpers, layout info
//   povar_lm.hip       the stages of an LM iteration: state, cost, linearise, prepare, apply (both steps), exports
//   povar_series.hip   the term loop: E0 / B^-1 launchers, kernel choice by timing, hipGraph, the series entry points
//   povar_comm.hip     exchange steps: RCCL / host-hook all-reduce, the peer-to-peer term exchange
//   povar_sc.hip       explicit-Schur-complement solvers (PCG / CHOLESKY / RIPCG)
// Device code: povar_kernels*.hpp (kernels that are not templates have internal linkage: every unit compiles what it launches).
#pragma once
#include "../../include/povar_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "povar_kernels.hpp"
#include "povar_kernels_joint.hpp"
#include "povar_kernels_sc.hpp"
#include "povar_kernels_chol.hpp"
#include "lpl_layout.hpp"
#include "ck_layout.hpp"
#include "povar_kernels_ck.hpp"
#include "povar_kernels_ck_det.hpp"
#include "povar_kernels_ck_joint.hpp"
#include "res_layout.hpp"
#include "povar_kernels_res.hpp"

using namespace povar;

inline thread_local std::string g_err;

inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(-(int)e_ - 1000, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

#define NCCL_TRY(expr)                                                                      \
  do {                                                                                      \
    ncclResult_t e_ = (expr);                                                               \
    if (e_ != ncclSuccess)                                                                  \
      return fail(-(int)e_ - 2000, std::string(#expr) + ": " + ncclGetErrorString(e_));     \
  } while (0)

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count, size_t* total) {
    n = count;
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess && total) *total += count * sizeof(T);
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
};

extern std::mutex g_capture_mu;  // see povar_ctx::placer_cancel (defined in povar_create.hip)

struct povar_ctx {
  int n_cams = 0, n_lms = 0;
  int64_t n_obs = 0;
  int n_bins = 0, n_slots = 0, n_items = 0, n_long = 0;
  int n_reg_blocks = 0, n_cam_blocks = 0;
  int n_hot = 0, n_hot_acc = 0, e0c_grid = 0, e0c_bins_per_wg = 0;
  int cu_limit = 0;              // CUs of the stream's mask (POVAR_CU_MASK), 0: the whole device
  int64_t n_cold = 0;
  int n_cold_items = 0;
  povar_options opt{};
  hipStream_t stream = nullptr;
  size_t bytes = 0;
  // camera-major landmark copies of the legacy kernels (cm_h and the cold views cc/c2) are built lazily in the
  // lane-per-landmark mode, which does not read them: lin_id counts linearisations, views_lin_id is the one they hold
  int64_t lin_id = 0, views_lin_id = -1, aux_lin_id = -1;  // aux: the per-slot sqrt(w) / weighted residual arrays
  // prepare_lpl[_h] writes only the lane-per-landmark records; hll_inv / lmrec of the legacy kernels follow lazily
  int64_t prep_id = 0, aux_prep_id = 0, prep_lin_id = -1;
  char* pin = nullptr;  // pinned host block of the small read-backs (read_scal_flags)
  size_t pin_bytes = 0;

  std::vector<int> slot_of_obs;  // host copy for exports in the reference's order
  std::vector<int> lm_off;

  // static
  DevBuf<double2> uv, cm_uv, tiles;
  DevBuf<int2> cc_cam_range;  // per camera: (first, end) position of its run in the cold camera-major view
  DevBuf<int> cold_pos;       // per slot: position in the cold view (-1: accumulated in LDS)
  // default mode with long landmarks: e0_lm_cached walks them itself (view "A" of the cold observations)
  DevBuf<int> c2_lm, c2_pos;
  DevBuf<int2> c2_range;
  DevBuf<double> c2_h;
  int64_t n_cold2 = 0;
  bool long_in_kernel = false;
  // lane-per-landmark layout of e0_lpl (struct V2)
  DevBuf<double2> v2_uv;
  DevBuf<int> v2_lm_of;
  DevBuf<int> v2_cw, v2_cpos, v2_lm_pos, v2_of_slot, v2_seg, v2_wg_tile_off, v2_wg_cam_off, v2_wg_cams, v2_wg_slot_rec, c3_lm;
  DevBuf<int2> v2_part_range, c3_range;
  DevBuf<int> c3_src, pl_c3_src;  // CmView::src of the lane-per-landmark cold view (row-major q: lpl_cold_q)
  // Cold observations of the lane-per-landmark kernels leave q row-major, side by side with the other lanes of their
  // row, and the per-camera kernels gather it -- instead of one scattered 32-byte store per lane into the camera-major
  // view (a third of the term on final-13682 with 24 % cold observations).  The gather costs the per-camera kernel a
  // dependent load, so graphs with fewer cold observations keep the direct store: from 20 % on (POVAR_COLD_Q_ROWS=0|1).
  // venice-1778 shape, cold share -> terms/s direct / row-major: Zipf(1) 3 % 14.8 / 14.3 k, 25 % long tracks 11 % 8.5 / 8.1 k,
  // Zipf(0.5) 18 % 9.76 / 9.77 k, uniform 31 % 6.46 / 7.11 k; final-13682 24 % 1.34 / 1.50 k.
  bool q_rows = false;
  DevBuf<double> c3_h, v2_part;
  int64_t n_cold3 = 0;
  int v2_max_slots = 0, v2_n_global = 0, v2_n_tail = 0, v2_strategy = 0;
  DevBuf<int4> v2_tile;
  DevBuf<double> v2_w, v2_lmrec;
  int64_t v2_rows = 0;
  // LDS bank placement of the rows (lpl_layout.hpp) on a host thread: povar_create returns on the natural row order,
  // the placed rows are swapped in at the next linearisation after they are ready (or by povar_layout_finalize).
  // POVAR_LPL_PLACE = sync | async | none; default: async from 2^20 observations on, sync below
  std::thread placer;
  std::atomic<int> placer_state{0};  // 0 no thread, 1 running, 2 rows uploaded and ready, 3 failed
  std::atomic<bool> placer_cancel{false};  // povar_destroy: do not finish a placement nobody will use
  // A hipMalloc / hipMemcpy of the host thread while the caller's thread captures the term loop into a hipGraph
  // invalidates the capture ("operation failed due to a previous error during capture", also in thread-local capture
  // mode): the thread makes its HIP calls in short pieces under g_capture_mu (one for the process: the check is not
  // per stream), the capture holds it from begin to end.
  DevBuf<double2> pl_uv;
  DevBuf<int> pl_cw, pl_cpos, pl_lm_pos, pl_lm_of, pl_of_slot;
  size_t pl_bytes = 0;
  int placement = 0;          // povar_layout_info::placement: 0 natural order, 1 placed in povar_create, 2 pending, 3 swapped in
  double placement_ms = 0;    // host wall time of the background build (valid from state 2 on)
  bool use_lpl = true;        // POVAR_E0_V1=1: keep e0_lm_cached<true> (lane per observation) for A/B runs
  bool lpl_forced = false;      // POVAR_E0_V1 set: keep the choice (the peer-to-peer exchange otherwise turns use_lpl on)
  bool use_lpl_prepare = true;  // POVAR_PREPARE_V1=1: keep lm_regular<OpPrepare> + cm_scatter
  // camera-chunk layout of e0_ck (ck_layout.hpp): derived from the lane-per-landmark rows in use, so it is rebuilt
  // with them (pl_ck: built by the placement thread from the placed rows, swapped in together with them)
  struct CkDev {
    DevBuf<double2> uv;
    DevBuf<int2> uvp;            // packed image points (CkLayout::uvp) instead of uv
    bool packed = false;
    DevBuf<int> cpos;            // cold-view positions of the entries of chunks without an accumulator slot (CkLayout::cpos)
    bool cold_q = false;         // ... which leave q there instead of partial records of their own
    DevBuf<uint32_t> li;
    DevBuf<int> src, bt_off, slot_rec;
    DevBuf<int2> lane_meta;
    DevBuf<int4> tile;
    DevBuf<int2> part_range;
    DevBuf<double> part, w;
    DevBuf<uint8_t> lcnt;        // e0_ck_det: ceil(log2(observations)) per landmark lane
    DevBuf<uint16_t> tick;       // e0_ck_det: ticket of every run total
    int nb = 0, slots = 0, n_part_rec = 0, max_acc = 0, max_tiles_bt = 0;
    int stride = 0;              // step 2: CKH_STRIDE or CKH_STRIDE_WIDE, whichever the layout was cut for (CkLayout::stride)
    int64_t n_capped_obs = 0;
    int64_t rows = 0, li_rows = 0, n_chunks = 0, n_cold_chunks = 0;
    double build_ms = 0;
    int64_t w_lin_id = -1;       // linearisation whose robust weights w holds
    bool ready = false;
    void release() {
      cpos.release(); cold_q = false;
      uv.release(); uvp.release(); li.release(); src.release(); lane_meta.release(); bt_off.release();
      slot_rec.release(); tile.release(); part_range.release(); part.release(); w.release();
      packed = false;
      lcnt.release(); tick.release();
      ready = false;
    }
  } ck, pl_ck,       // step 1 (e0_ck): the layout in use / the one the placement thread built for the placed rows
    ckh, pl_ckh;     // step 2 (e0_ck_h): a second instance (64 bytes of LDS per landmark slot: more batches, other chunks)
  // resident power series (series_res, povar_kernels_res.hpp): the layout of res_layout.hpp on the device
  struct ResDev {
    DevBuf<int> lane_cam, lane_seg, lslot, oslot, wave_h, lm_off, lm_id, cam_off, cam_id, cam_zi, own_off, own_cam, own_zi, oq_off, oq_rec;
    DevBuf<double2> uv;
    DevBuf<int2> own_q;
    DevBuf<uint4> part, zbuf, nrm;   // granule pairs (povar_kernels_res.hpp)
    DevBuf<unsigned> launch;         // launch counter: the high bits of the granule tags
    int W = 0, NW = 0, H = 0, R = 1, LS = 1, n_rec = 0, max_lm = 0, max_cam = 0, max_oq = 0, max_own = 0, max_chunks = 0, order = 0;
    size_t lds_bytes = 0;
    double build_ms = 0;
    bool ready = false;
    void release() {
      lane_cam.release(); lane_seg.release(); lslot.release(); oslot.release(); wave_h.release();
      lm_off.release(); lm_id.release(); cam_off.release(); cam_id.release(); cam_zi.release(); own_off.release(); own_cam.release();
      own_zi.release(); oq_off.release(); oq_rec.release();
      uv.release(); own_q.release(); part.release(); zbuf.release(); nrm.release(); launch.release();
      ready = false;
    }
  } res;
  int res_mode = -1;             // -1: the library times the resident series against the per-term kernels once per context
                                 // (res_autotune); 0: per-term kernels; 1: resident series whenever the context allows
  bool res_tuned = false, res_choice = false, res_failed = false;
  float res_tune_us[2] = {0, 0}; // per term: per-term kernels (hipGraph), resident series
  bool res_check = false;        // a resident series is in flight whose give-up bit (flags[0] & 4) has not been looked at
  int res_last_m = 0;
  double res_last_tol[2] = {0, 0};
  unsigned res_spin_limit = 1u << 18;
  bool deterministic = false;    // POVAR_DETERMINISTIC=1: the E0 mode and the kernel choices are pinned
  bool det_ck = false;           // ... and step 1's terms run e0_ck_det where the chunk layout fits (else: the gather form)
  bool det_check = false;        // a series of e0_ck_det is in flight whose failure bit (flags[0] & 8) has not been looked at
  DevBuf<int2> ck_zero_range;    // [n_cams] empty runs: e0_ck leaves no per-observation cold view to the per-camera kernels
  int ck_variant = 0;            // 0: e0_lpl; 1..CK_VARIANTS: e0_ck instantiation (POVAR_CK_VARIANTS)
  int ckh_variant = 0;           // step 2: 0: e0_lpl_h; 1: e0_ck_h
  bool ckh_tuned = false;
  float ckh_tune_us[2] = {0, 0}; // e0_lpl_h, e0_ck_h
  bool ck_auto = true;           // the library picks e0_lpl or e0_ck by timing both on this problem (ck_autotune); false: forced
  bool ck_tuned = false;
  bool ck_fresh[2] = {false, false};  // a timing of step 1 / step 2 finished on this rank since the ranks last agreed (tune_agree)
  float ck_tune_us[2] = {0, 0};  // what the timing saw: e0_lpl, e0_ck (microseconds per launch)
  DevBuf<double4> q4c;        // scatter scalars of the cold observations, in cold camera-major order
  DevBuf<int> lm_slot0, lm_cnt_dev;
  bool k1_qr = true;          // POVAR_K1_NORMAL_EQ=1: the round-1 normal-equation kernels (A/B accuracy runs)
  // stage timings (povar_timings): hipEvent pairs around the entry points, summed on demand
  bool timings_on = false;
  std::vector<hipEvent_t> tev;
  std::vector<int> tev_kind;
  size_t tev_used = 0;
  povar_timings_info tsum{};
  DevBuf<int> cam, lm, meta, hot_cams, cam_hot, cc_slot, cc_lm, cc_item_off, cc_cam_item_off, long_lm, long_first, long_cnt, cm_slot, cm_lm, item_off, item_cam,
      cam_item_off, flags;
  // state
  DevBuf<double4> cams4, cams_lin4, cams_bak4, lms4, lms_lin4, lms_bak4, jl_scale4, rres, q4;
  DevBuf<double> hll_inv, sw, sigma, diag2, G, binv, b, tmp, accum, z, y, inc, item_part,
      item_partG, norm_part, norms, part, scal, stage, cm_h, lmrec, ncw, cc_h, cc_part, hot_part, hot_rec, zimg;

  // explicit-SC solvers (PCG / CHOLESKY / RIPCG), allocated on first use
  DevBuf<double> sc_dm_part, sc_dm, sc_bmat, sc_minv, sc_x, sc_r, sc_p, sc_q, sc_zv, sc_part, sc_s;
  ScP sc{};
  DevBuf<double> sc_dense, sc_xpad;      // CHOLESKY: augmented S (povar_kernels_chol.hpp) and the padded solution
  DevBuf<int> sc_lm_slot0, sc_lm_cnt, sc_info;

  Dp d{};
  bool new_linearization_point = false;  // linearizor_power_varproj.cpp:75, 192, 240
  bool linearized = false;
  bool tiles_valid = false;
  bool joint = false;        // system prepared last: step 2 (11-dim tangent) or step 1 (12-dim)
  bool linearized_h = false;
  double alpha_lin = 0;

  // multi-GPU
  // peer-to-peer term exchange (povar_p2p_export / povar_p2p_attach)
  double* xbuf = nullptr;              // this rank's exchange buffer [2][world][n_cams][16]
  size_t xbuf_count = 0;
  std::vector<double*> peer_host;      // opened peer mappings (index = rank; own entry = xbuf)
  DevBuf<double*> peer_dev;
  DevBuf<unsigned long long> p2p_epoch;
  bool p2p = false;
  ncclComm_t comm = nullptr;
  povar_allreduce_fn host_fn = nullptr;  // caller-supplied exchange (povar_comm_init_host)
  void* host_user = nullptr;
  std::vector<double> host_stage;
  int world = 1, rank = 0;

  // hipGraph of the m-term series loop (launch-bound on small problems and at 8 GPUs)
  double create_ms = 0;  // host wall time of povar_create (layout construction + uploads)
  // lane-ordered mirrors of lms4 / lms_lin4 / jl_scale4 (V2::lmx, lml, lsc): lms_ver counts the writes to lms4, the
  // *_ver / *_lin_id fields say what each mirror currently reflects
  DevBuf<double4> v2_lmx, v2_lml, v2_lsc;
  uint64_t lms_ver = 1, lmx_ver = 0;
  int64_t lml_lin_id = -1, lsc_lin_id = -1;
  int64_t jls_lin_id = -1;  // linearisation whose Jl column scale the landmark-order master jl_scale4 holds
  int64_t lmslin_lin_id = -1;  // ... and whose landmarks the landmark-order master lms_lin4 holds (lazily, from lml)
  // compute_error_* of an unchanged state (the LM loop asks again at the top of every iteration,
  // bal_bundle_adjustment.cpp:302-310 / 600-605): cams_ver counts the writes to cams4 as lms_ver does for lms4
  uint64_t cams_ver = 1;
  struct ErrMemo {
    bool valid = false;
    uint64_t lms_ver = 0, cams_ver = 0;
    double alpha = 0;
    int kind = 0, mode = 0;
    povar_residual_info ri{};
  } err_memo;
  bool no_err_memo = false;   // POVAR_NO_ERR_MEMO=1: evaluate every compute_error call (timing the kernel itself)
  bool has_empty_lm = false;  // a landmark without observations has no lane: lms_lin4 is then copied eagerly
  bool flag0_clean = false;  // flags[0] is known to be zero on the device (read back as zero, no writer launched since)
  hipGraphExec_t series_graph = nullptr;
  Dp series_graph_d{};
  int series_graph_key[6] = {0, 0, 0, 0, 0, 0};
  double series_graph_tol[2] = {0, 0};
  bool use_graph = true;
  bool graph_with_comm = false;
  bool fuse_binv = true;   // POVAR_NO_FUSE=1: keep cam_cold_sum and cam_binv_axpy separate

  // profiling
  bool profile = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_kind;  // 0 e0, 1 binv, 2 comm  (interval between ev[i], ev[i+1])
  size_t ev_used = 0;
};

// e0_ck instantiations (povar_ctx::ck_variant): wavefronts per workgroup, rows a tile keeps in flight, double-buffered
// tile records, groups of wavefronts working on different batches (povar_kernels_ck.hpp)
#define POVAR_CK_VARIANTS(X) \
  X(1, 16, 2, false, 1) X(2, 16, 4, false, 1) X(3, 12, 2, true, 1) X(4, 16, 2, false, 2) X(5, 16, 4, false, 2) X(6, 8, 2, true, 1)
constexpr int CK_VARIANTS = 6;
struct CkVariant { int nw, sd; bool db; int ng; };
inline CkVariant ck_variant_info(int variant) {
  switch (variant) {
#define X(id, nw, sd, db, ng) case id: return CkVariant{nw, sd, db, ng};
    POVAR_CK_VARIANTS(X)
#undef X
    default: return CkVariant{16, 2, false, 1};
  }
}


template <class T>
int upload(DevBuf<T>& buf, const std::vector<T>& v, povar_ctx* c) {
  HIP_TRY(buf.alloc(std::max<size_t>(v.size(), 1), &c->bytes));
  if (!v.empty()) HIP_TRY(hipMemcpy(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

inline int grid_for(int64_t n, int block) { return (int)((n + block - 1) / block); }


// ------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------
template <class Op>
void launch_lm(povar_ctx* c, const Op& op) {
  hipLaunchKernelGGL((lm_regular<Op>), dim3(c->n_reg_blocks), dim3(LM_BLOCK), 0, c->stream, c->d, op,
                     c->part.p);
  if (c->n_long > 0)
    hipLaunchKernelGGL((lm_long<Op>), dim3(c->n_long), dim3(LM_BLOCK), 0, c->stream, c->d, op, c->part.p);
}

template <int N>
void launch_reduce(povar_ctx* c, double* out) {
  hipLaunchKernelGGL((reduce_partials<N>), dim3(1), dim3(1024), 0, c->stream, c->part.p,
                     c->n_reg_blocks + c->n_long, out);
}

inline bool sharded(const povar_ctx* c) { return c->comm != nullptr || c->host_fn != nullptr; }

// the per-term exchange runs through the peer-to-peer kernels: no library / host call inside the term loop

// povar_timings: a pair of events on the context's stream around an entry point (kinds: 0 linearize, 1 prepare,
// 2 solve = the power series / PCG / CHOLESKY, 3 apply = camera update + back-substitution, 4 other)
struct TimeScope {
  povar_ctx* c;
  bool on;
  TimeScope(povar_ctx* c_, int kind) : c(c_), on(c_->timings_on) {
    if (!on) return;
    if (c->tev_used + 2 > c->tev.size()) {
      for (int i = 0; i < 2; ++i) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        c->tev.push_back(e);
        c->tev_kind.push_back(-1);
      }
    }
    c->tev_kind[c->tev_used] = kind;
    (void)hipEventRecord(c->tev[c->tev_used], c->stream);
  }
  ~TimeScope() {
    if (!on) return;
    (void)hipEventRecord(c->tev[c->tev_used + 1], c->stream);
    c->tev_used += 2;
  }
};

// hipEvents that are destroyed on every return path (the HIP_TRY early returns of the timing functions included)
template <int N>
struct EventSet {
  hipEvent_t e[N] = {};
  hipError_t create() {
    for (auto& x : e) {
      hipError_t r = hipEventCreate(&x);
      if (r != hipSuccess) return r;
    }
    return hipSuccess;
  }
  hipEvent_t& operator[](int i) { return e[i]; }
  ~EventSet() { for (auto& x : e) if (x) (void)hipEventDestroy(x); }
};


// ---- functions one translation unit defines and another calls
CkP ck_params(const povar_ctx* c, const povar_ctx::CkDev& D);  // povar_series.hip
CkP ck_params(const povar_ctx* c);  // povar_series.hip
bool ck_det_possible(const povar_ctx* c);  // povar_series.hip
bool ck_det_active(const povar_ctx* c);  // povar_series.hip
bool ck_active(const povar_ctx* c);  // povar_series.hip
bool ckh_det_possible(const povar_ctx* c);  // povar_series.hip
bool ckh_active(const povar_ctx* c);  // povar_series.hip
void ck_dp(const povar_ctx* c, Dp& da);  // povar_series.hip
bool ck_variant_fits(const povar_ctx* c, int variant);  // povar_series.hip
void launch_e0_ck(povar_ctx* c, const Dp& da);  // povar_series.hip
hipError_t ck_set_lds_all();  // povar_series.hip
void ensure_ck_w(povar_ctx* c);  // povar_series.hip
void launch_e0_ck_h(povar_ctx* c, const Dp& da);  // povar_series.hip
bool res_variant_exists(int nw, int h, int rr, int ls);  // povar_series.hip
void launch_res(povar_ctx* c, const ResP& k);  // povar_series.hip
hipError_t res_set_lds_all();  // povar_series.hip
bool res_possible(const povar_ctx* c);  // povar_series.hip
bool res_active(const povar_ctx* c);  // povar_series.hip
ResP res_params(const povar_ctx* c, int m, double q_tol, double r_tol);  // povar_series.hip
int enqueue_series_res(povar_ctx* c, int32_t m, double q_tol, double r_tol);  // povar_series.hip
void prof_mark(povar_ctx* c, int kind);  // povar_series.hip
Dp ldsacc_dp(povar_ctx* c, bool long_in_kernel = false);  // povar_series.hip
int launch_e0(povar_ctx* c, int* binv_mode, int fuse_norms = -1);  // povar_series.hip
void launch_binv(povar_ctx* c, int mode, int want_norms);  // povar_series.hip
int ck_autotune(povar_ctx* c);  // povar_series.hip
int tune_agree(povar_ctx* c, int step);  // povar_series.hip
int ckh_autotune(povar_ctx* c);  // povar_series.hip
extern "C" int enqueue_series(povar_ctx* c, int32_t m, double q_tol, double r_tol);  // povar_series.hip
extern "C" int run_series(povar_ctx* c, int32_t m, double q_tol, double r_tol, bool use_res);  // povar_series.hip
extern "C" int res_verify(povar_ctx* c);  // povar_series.hip
extern "C" int res_autotune(povar_ctx* c, int32_t m, double q_tol, double r_tol);  // povar_series.hip
int ck_max_cams();  // povar_create.hip
bool ck_upload(povar_ctx* c, povar_ctx::CkDev& D, const CkLayout& K, bool locked, size_t* bytes, bool need_uv = true);  // povar_create.hip
int res_upload(povar_ctx* c, const ResLayout& R);  // povar_create.hip
void res_build_for(int n_cams, int n_lms, const int32_t* lm_off, const int32_t* cam_idx, const double* obs, const std::vector<int>& rank1, const std::vector<int>& slot_of_obs, int wgs, ResLayout& R);  // povar_create.hip
int swap_in_placed_rows(povar_ctx* c, bool wait);  // povar_create.hip
int check_ctx(povar_ctx* c);  // povar_create.hip
int ensure_pin(povar_ctx* c);  // povar_create.hip
int read_scal_flags(povar_ctx* c, double* h, int n, int (&f)[4]);  // povar_create.hip
int read_flags(povar_ctx* c, int (&f)[4]);  // povar_create.hip
int read_scal(povar_ctx* c, double* h, int n);  // povar_create.hip
int write_cam_vector(povar_ctx* c, double* dst, const double* in, size_t n);  // povar_create.hip
int read_cam_vector(povar_ctx* c, double* out, const double* src, size_t n);  // povar_create.hip
bool lpl_only(const povar_ctx* c);  // povar_lm.hip
void build_views(povar_ctx* c);  // povar_lm.hip
void lanes_from(povar_ctx* c, const double4* src, double4* dst);  // povar_lm.hip
void ensure_lmx(povar_ctx* c);  // povar_lm.hip
void ensure_lin_mirrors(povar_ctx* c);  // povar_lm.hip
void ensure_jl_scale4(povar_ctx* c);  // povar_lm.hip
void ensure_lms_lin(povar_ctx* c);  // povar_lm.hip
void ensure_legacy(povar_ctx* c);  // povar_lm.hip
int clear_flag0(povar_ctx* c);  // povar_lm.hip
bool err_memo_hit(const povar_ctx* c, int kind, double alpha, povar_residual_info* out);  // povar_lm.hip
void err_memo_store(povar_ctx* c, int kind, double alpha, const povar_residual_info& ri);  // povar_lm.hip
int combine_flag(povar_ctx* c, int* flag);  // povar_lm.hip
int ensure_tiles(povar_ctx* c);  // povar_lm.hip
extern "C" void set_alpha(povar_ctx* c, double alpha);  // povar_lm.hip
int allreduce(povar_ctx* c, double* buf, size_t n);  // povar_comm.hip
void p2p_dp(povar_ctx* c, Dp& dt);  // povar_comm.hip
int ensure_sc(povar_ctx* c);  // povar_sc.hip
int e0_dense(povar_ctx* c);  // povar_sc.hip
int run_cholesky(povar_ctx* c, int32_t* num_iterations, int32_t* termination);  // povar_sc.hip
