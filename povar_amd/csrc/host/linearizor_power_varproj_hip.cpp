// LinearizorPowerVarproj (solver/linearizor_power_varproj.cpp:21-308 + solver/linearizor_base.cpp:48-100)
// and LinearizorSC (solver/linearizor_sc.cpp:50-360; PCG / CHOLESKY / RIPCG) on top of the C ABI of
// include/povar_hip.h: the two reference classes differ in solve()/solve_joint() and in the Jl column
// scaling of step 1 only, so one class serves both.  The device context owns cameras and landmarks
// between calls; BalProblem forwards backup/restore/normalise through StateMirror.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <memory>
#include <cstring>
#include <cstdlib>

#include "../../../include/povar_hip.h"
#include "linearizor.hpp"

namespace povar_host {

namespace {

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  double reset() { const double e = elapsed(); t0 = std::chrono::steady_clock::now(); return e; }
};

#define IF_SET(P) if (P) P

void check(int rc, const char* what) {
  if (rc < 0) {  // hard failures abort like CHECK / LOG(FATAL) (linearizor_power_varproj.cpp:59-60)
    std::fprintf(stderr, "FATAL: %s failed (%d): %s\n", what, rc, povar_last_error());
    std::abort();
  }
}

// what a device context is built from: sizes, options, every (camera, u, v) in landmark order
std::string context_key(const BalProblem& p, const povar_options& o) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
  long n_obs = 0;
  for (const auto& lm : p.landmarks()) {
    mix(lm.obs.size());
    for (const auto& kv : lm.obs) {
      unsigned long long bu, bv;
      std::memcpy(&bu, &kv.second[0], 8);
      std::memcpy(&bv, &kv.second[1], 8);
      mix((unsigned long long)kv.first);
      mix(bu);
      mix(bv);
    }
    n_obs += (long)lm.obs.size();
  }
  char buf[256];
  std::snprintf(buf, sizeof buf, "%d/%d/%ld/%016llx/norm%d/%.17g/%.17g/dev%d/e0%d/f%x", p.num_cameras(), p.num_landmarks(), n_obs, h,
                o.robust_norm, o.huber_parameter, o.jacobi_scaling_eps, o.device, o.e0_mode, o.flags);
  return buf;
}

ResidualInfo to_ri(const povar_residual_info& r) {
  ResidualInfo o;
  o.all = {r.all_num_obs, r.all_error, r.all_residual_sum};
  o.valid = {r.valid_num_obs, r.valid_error, r.valid_residual_sum};
  o.is_numerically_valid = r.is_numerically_valid != 0;
  return o;
}

class LinearizorPowerVarprojHip : public Linearizor, public StateMirror {
 public:
  LinearizorPowerVarprojHip(BalProblem& bal_problem, const SolverOptions& options, SolverSummary* summary,
                            bool homogeneous)
      : options_(options), bal_problem_(bal_problem), summary_(summary), homogeneous_(homogeneous) {
    povar_options o{};
    o.robust_norm = (int)options.residual.robust_norm;
    o.huber_parameter = options.residual.huber_parameter;
    // get_effective_jacobi_scaling_epsilon, linearizor_base.cpp:94-100 (Sophus epsilonSqrt<double> = 1e-5)
    o.jacobi_scaling_eps = options.jacobi_scaling_epsilon > 0 ? options.jacobi_scaling_epsilon : 1e-5;
    o.device = options.device;
    o.e0_mode = options.e0_mode == "tiles" ? POVAR_E0_TILES
                : options.e0_mode == "implicit" ? POVAR_E0_IMPLICIT : POVAR_E0_IMPLICIT_LDSACC;
    o.flags = options.deterministic ? POVAR_FLAG_DETERMINISTIC : 0u;
    // The reference builds a new linearizor for step 2 (its constructor only allocates: sc/linearization_varproj.hpp:
    // 44-60); the device context -- observation layout, camera sets, row placement under way -- depends on the
    // observations and these options only, so the one step 1 left behind is taken over.
    cache_key_ = context_key(bal_problem, o);
    if (bal_problem.device_cache.ctx && bal_problem.device_cache.key == cache_key_) {
      holder_ = std::move(bal_problem.device_cache.ctx);
      bal_problem.device_cache = BalProblem::DeviceCache();
      ctx_ = static_cast<povar_ctx*>(holder_.get());
    } else {
      bal_problem.device_cache = BalProblem::DeviceCache();  // a context for something else: release it first
      std::vector<int> lm_off, cam_idx;
      std::vector<double> obs;
      bal_problem.flatten(lm_off, cam_idx, obs);
      check(povar_create(&ctx_, bal_problem.num_cameras(), bal_problem.num_landmarks(), (int64_t)cam_idx.size(),
                         lm_off.data(), cam_idx.data(), obs.data(), &o), "povar_create");
      holder_ = std::shared_ptr<void>(ctx_, [](void* c) { povar_destroy(static_cast<povar_ctx*>(c)); });
    }
    using ST = SolverOptions::SolverType;
    sc_step1_ = !homogeneous && (options.solver_type_step_1 == ST::PCG || options.solver_type_step_1 == ST::CHOLESKY);
    sc_step2_ = homogeneous && options.solver_type_step_2 == SolverOptions::SolverTypeRiemannian::RIPCG;
    // LinearizorSC::linearize_pOSE does not scale the landmark Jacobian columns (linearizor_sc.cpp:163-191)
    check(povar_set_jl_col_scaling(ctx_, sc_step1_ ? 0 : 1), "povar_set_jl_col_scaling");
    // IterationSummary timings come from the device (hipEvents on the library's stream, povar_timings), not from
    // host clocks around the calls
    check(povar_timings_enable(ctx_, 1), "povar_timings_enable");
    push_state();
    bal_problem_.mirror = this;
  }
  ~LinearizorPowerVarprojHip() override {
    pull_state();
    bal_problem_.mirror = nullptr;
    bal_problem_.device_cache.ctx = std::move(holder_);  // destroyed with the problem, or taken over by the next linearizor
    bal_problem_.device_cache.key = cache_key_;
  }

  void start_iteration(IterationSummary* it) override { it_summary_ = it; }
  void finish_iteration() override {}

  void initialize_varproj_lm_pOSE(double alpha, bool initialization_varproj) override {
    if (initialization_varproj) check(povar_init_landmarks_pose(ctx_, alpha), "povar_init_landmarks_pose");
  }
  void compute_error_pOSE(ResidualInfo& ri, bool) override {
    Timer t;
    povar_residual_info r;
    check(povar_error_pose(ctx_, options_.alpha, &r), "povar_error_pose");
    ri = to_ri(r);
    (void)t;
    IF_SET(it_summary_)->residual_evaluation_time_in_seconds += device_seconds(4);
    IF_SET(summary_)->num_residual_evaluations += 1;
  }
  void compute_error_homogeneous(ResidualInfo& ri, bool) override {
    Timer t;
    povar_residual_info r;
    check(povar_error_homogeneous(ctx_, &r), "povar_error_homogeneous");
    ri = to_ri(r);
    (void)t;
    IF_SET(it_summary_)->residual_evaluation_time_in_seconds += device_seconds(4);
    IF_SET(summary_)->num_residual_evaluations += 1;
  }
  void linearize_pOSE(double alpha) override {
    Timer t;
    const int rc = povar_linearize_pose(ctx_, alpha);
    check(rc, "povar_linearize_pose");
    if (rc == POVAR_NUMERIC_FAILURE) {
      std::fprintf(stderr, "FATAL: did not expect numerical failure during linearization\n");
      std::abort();
    }
    (void)t;
    const double e = device_seconds(0);
    IF_SET(it_summary_)->jacobian_evaluation_time_in_seconds = e;
    IF_SET(it_summary_)->stage1_time_in_seconds = e;
    IF_SET(summary_)->num_jacobian_evaluations += 1;
  }
  void linearize_projective_space_homogeneous() override {
    Timer t;
    const int rc = povar_linearize_homogeneous(ctx_);
    check(rc, "povar_linearize_homogeneous");
    if (rc == POVAR_NUMERIC_FAILURE) {
      std::fprintf(stderr, "FATAL: did not expect numerical failure during linearization\n");
      std::abort();
    }
    (void)t;
    const double e = device_seconds(0);
    IF_SET(it_summary_)->jacobian_evaluation_time_in_seconds = e;
    IF_SET(it_summary_)->stage1_time_in_seconds = e;
    IF_SET(summary_)->num_jacobian_evaluations += 1;
  }
  VecX solve(const SolverOptions& so, double lambda, double) override {
    VecX inc(12 * (size_t)bal_problem_.num_cameras());
    Timer t;
    if (sc_step1_) {  // LinearizorSC::solve, linearizor_sc.cpp:85-160
      const bool chol = so.solver_type_step_1 == SolverOptions::SolverType::CHOLESKY;
      if (!chol) require_schur_jacobi();
      int32_t iters = 0, term = 0;
      check(povar_solve_pose_sc(ctx_, lambda, chol ? POVAR_SC_CHOLESKY : POVAR_SC_PCG, options_.min_linear_solver_iterations,
                                options_.max_linear_solver_iterations, options_.eta, inc.data(), &iters, &term),
            "povar_solve_pose_sc");
      IF_SET(it_summary_)->prepare_time_in_seconds = device_seconds(1);
      fill_sc_summary(device_seconds(2), iters, term, chol);
      return inc;
    }
    const int st = so.solver_type_step_1 == SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT
                       ? POVAR_POWER_SCHUR_COMPLEMENT : POVAR_POWER_VARPROJ;
    check(povar_prepare_pose(ctx_, lambda, st), "povar_prepare_pose");
    int32_t iters = 0, term = 0;
    check(povar_power_series_pose(ctx_, options_.power_sc_iterations, options_.eta, options_.r_tolerance, &iters, &term),
          "povar_power_series_pose");
    check(povar_get_increment(ctx_, inc.data()), "povar_get_increment");
    (void)t;
    IF_SET(it_summary_)->prepare_time_in_seconds = device_seconds(1);
    fill_solver_summary(device_seconds(2), iters, term);
    return inc;
  }
  VecX solve_joint(double lambda, double) override {
    VecX inc(11 * (size_t)bal_problem_.num_cameras());
    Timer t;
    int32_t iters = 0, term = 0;
    if (sc_step2_) {  // LinearizorSC::solve_joint, linearizor_sc.cpp:224-303
      require_schur_jacobi();
      check(povar_solve_joint_sc(ctx_, lambda, options_.min_linear_solver_iterations, options_.max_linear_solver_iterations,
                                 options_.eta, inc.data(), &iters, &term), "povar_solve_joint_sc");
      IF_SET(it_summary_)->prepare_time_in_seconds = device_seconds(1);
      fill_sc_summary(device_seconds(2), iters, term, false);
      return inc;
    }
    check(povar_solve_joint(ctx_, lambda, options_.power_sc_iterations, options_.eta, options_.r_tolerance,
                            inc.data(), &iters, &term), "povar_solve_joint");
    (void)t;
    IF_SET(it_summary_)->prepare_time_in_seconds = device_seconds(1);
    fill_solver_summary(device_seconds(2), iters, term);
    return inc;
  }
  double apply(const SolverOptions& so, double alpha, VecX&& inc) override {
    Timer t;
    double l_diff = 0;
    const int st = so.solver_type_step_1 == SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT
                       ? POVAR_POWER_SCHUR_COMPLEMENT : POVAR_POWER_VARPROJ;
    check(povar_apply_pose(ctx_, st, alpha, inc.data(), &l_diff), "povar_apply_pose");
    (void)t;
    IF_SET(it_summary_)->back_substitution_time_in_seconds = device_seconds(3);
    return l_diff;
  }
  double apply_joint(VecX&& inc) override {
    Timer t;
    double l_diff = 0;
    check(povar_apply_joint(ctx_, inc.data(), &l_diff), "povar_apply_joint");
    (void)t;
    IF_SET(it_summary_)->back_substitution_time_in_seconds = device_seconds(3);
    return l_diff;
  }

  // StateMirror
  void backup_pOSE() override { check(povar_backup_pose(ctx_), "povar_backup_pose"); }
  void restore_pOSE() override { check(povar_restore_pose(ctx_), "povar_restore_pose"); }
  void backup_joint() override { check(povar_backup_joint(ctx_), "povar_backup_joint"); }
  void restore_joint() override { check(povar_restore_joint(ctx_), "povar_restore_joint"); }
  void normalize_joint() override { check(povar_normalize_joint(ctx_), "povar_normalize_joint"); }
  void pull_state() override {
    const int nc = bal_problem_.num_cameras(), nl = bal_problem_.num_landmarks();
    std::vector<double> cams(12 * (size_t)nc), lms((homogeneous_ ? 4 : 3) * (size_t)nl);
    check(povar_get_cameras(ctx_, cams.data()), "povar_get_cameras");
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 12; ++k) bal_problem_.cameras()[c].space_matrix[k] = cams[12 * (size_t)c + k];
    if (homogeneous_) {
      check(povar_get_landmarks_homogeneous(ctx_, lms.data()), "povar_get_landmarks_homogeneous");
      for (int l = 0; l < nl; ++l)
        for (int k = 0; k < 4; ++k) bal_problem_.landmarks()[l].p_w_homogeneous[k] = lms[4 * (size_t)l + k];
    } else {
      check(povar_get_landmarks(ctx_, lms.data()), "povar_get_landmarks");
      for (int l = 0; l < nl; ++l)
        for (int k = 0; k < 3; ++k) bal_problem_.landmarks()[l].p_w[k] = lms[3 * (size_t)l + k];
    }
  }

 private:
  // seconds of device time the entry points of `kind` have taken since the previous call of this helper
  // (0 linearize, 1 prepare, 2 solve, 3 apply, 4 other)
  double device_seconds(int kind) {
    povar_timings_info t;
    check(povar_timings(ctx_, &t), "povar_timings");
    const double now[5] = {t.linearize_ms, t.prepare_ms, t.solve_ms, t.apply_ms, t.other_ms};
    const double d = (now[kind] - seen_[kind]) * 1e-3;
    seen_[kind] = now[kind];
    return d;
  }
  double seen_[5] = {0, 0, 0, 0, 0};
  void push_state() {
    const int nc = bal_problem_.num_cameras(), nl = bal_problem_.num_landmarks();
    std::vector<double> cams(12 * (size_t)nc), lms((homogeneous_ ? 4 : 3) * (size_t)nl);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 12; ++k) cams[12 * (size_t)c + k] = bal_problem_.cameras()[c].space_matrix[k];
    check(povar_set_cameras(ctx_, cams.data()), "povar_set_cameras");
    if (homogeneous_) {
      for (int l = 0; l < nl; ++l)
        for (int k = 0; k < 4; ++k) lms[4 * (size_t)l + k] = bal_problem_.landmarks()[l].p_w_homogeneous[k];
      check(povar_set_landmarks_homogeneous(ctx_, lms.data()), "povar_set_landmarks_homogeneous");
    } else {
      for (int l = 0; l < nl; ++l)
        for (int k = 0; k < 3; ++k) lms[3 * (size_t)l + k] = bal_problem_.landmarks()[l].p_w[k];
      check(povar_set_landmarks(ctx_, lms.data()), "povar_set_landmarks");
    }
  }
  void fill_solver_summary(double seconds, int iters, int term) {
    IF_SET(it_summary_)->solve_reduced_system_time_in_seconds = seconds;
    IF_SET(it_summary_)->linear_solver_iterations = iters;
    // messages of linearization_power_varproj.hpp:213-235
    IF_SET(it_summary_)->linear_solver_message =
        term == POVAR_LINEAR_SOLVER_SUCCESS ? "Iteration: " + std::to_string(iters) + " Convergence."
                                            : "Maximum number of iterations reached.";
    IF_SET(it_summary_)->linear_solver_type = "bal_power_sc";
    IF_SET(summary_)->num_linear_solves += 1;
  }

  void require_schur_jacobi() const {  // CHECK of linearizor_sc.cpp:126-128, 268-270
    if (options_.preconditioner_type != SolverOptions::PreconditionerType::SCHUR_JACOBI) {
      std::fprintf(stderr, "FATAL: preconditioner_type: not implemented\n");
      std::abort();
    }
  }
  void fill_sc_summary(double seconds, int iters, int term, bool direct) {
    IF_SET(it_summary_)->solve_reduced_system_time_in_seconds = seconds;
    IF_SET(it_summary_)->linear_solver_iterations = iters;
    // messages of conjugate_gradient.hpp:114-301 (without the numbers they embed); solve_direct_pOSE
    // leaves the message empty (linearization_sc.hpp:239)
    IF_SET(it_summary_)->linear_solver_message =
        direct ? ""
        : term == POVAR_LINEAR_SOLVER_SUCCESS ? "Iteration: " + std::to_string(iters) + " Convergence."
        : term == POVAR_LINEAR_SOLVER_FAILURE ? "Numerical failure."
                                              : "Maximum number of iterations reached.";
    IF_SET(it_summary_)->linear_solver_type = "bal_sc";
    IF_SET(summary_)->num_linear_solves += 1;
  }

  SolverOptions options_;
  bool sc_step1_ = false, sc_step2_ = false;
  BalProblem& bal_problem_;
  SolverSummary* summary_ = nullptr;
  IterationSummary* it_summary_ = nullptr;
  povar_ctx* ctx_ = nullptr;
  std::shared_ptr<void> holder_;  // owns ctx_; handed to BalProblem::device_cache by the destructor
  std::string cache_key_;
  bool homogeneous_;
};

// ------------------------------------------------------------------------------------------
// N landmark shards in one process (BASELINE configs 4 / 5; SURVEY 8(b): "one context per process drives all local
// GPUs").  The reference's caller is ONE thread calling the eleven virtuals (solver/linearizor.hpp:65-81,
// bal_bundle_adjustment.cpp:283-284, 585-586); the library's sharded entry points are collectives -- every rank must be
// inside the same call at the same time -- so each shard context has a host thread of its own and every virtual hands
// the same call to all of them and waits.  Cameras and every per-camera result are replicated (identical on all ranks
// after the library's exchange steps: rank 0's copy is returned), landmarks live on the shard that owns them.
// Exchange: RCCL (povar_comm_init) when the shards sit on distinct devices; an in-process all-reduce through the
// library's host hook (povar_comm_init_host) when they share one (RCCL refuses duplicate devices) or POVAR_HOST_COMM=1.
// ------------------------------------------------------------------------------------------
class ShardTeam {
 public:
  explicit ShardTeam(int n) : n_(n), slab_(n) {
    for (int r = 0; r < n; ++r) workers_.emplace_back([this, r] { loop(r); });
  }
  ~ShardTeam() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  int size() const { return n_; }
  // fn(rank) on every shard's thread, concurrently; returns when all are done
  void run(const std::function<void(int)>& fn) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      pending_ = n_;
      ++gen_;
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }
  // in-process all-reduce (sum, in place, fixed rank order on every rank): the host hook of povar_comm_init_host
  static void allreduce_hook(double* buf, int64_t n, void* user) {
    auto* a = static_cast<std::pair<ShardTeam*, int>*>(user);
    a->first->allreduce(a->second, buf, n);
  }
  std::vector<std::pair<ShardTeam*, int>> hook_args;

 private:
  void loop(int r) {
    long seen = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (quit_) return;
        fn = fn_;
      }
      (*fn)(r);
      {
        std::lock_guard<std::mutex> lk(mu_);
        --pending_;
      }
      done_cv_.notify_all();
    }
  }
  void barrier() {
    std::unique_lock<std::mutex> lk(bar_mu_);
    const long g = bar_gen_;
    if (++bar_count_ == n_) {
      bar_count_ = 0;
      ++bar_gen_;
      bar_cv_.notify_all();
    } else {
      bar_cv_.wait(lk, [&] { return bar_gen_ != g; });
    }
  }
  void allreduce(int rank, double* buf, int64_t n) {
    slab_[rank].assign(buf, buf + n);
    barrier();
    for (int64_t i = 0; i < n; ++i) {
      double s = 0;
      for (int r = 0; r < n_; ++r) s += slab_[r][i];
      buf[i] = s;
    }
    barrier();
  }
  int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_, bar_mu_;
  std::condition_variable cv_, done_cv_, bar_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int pending_ = 0, bar_count_ = 0;
  long gen_ = 0, bar_gen_ = 0;
  bool quit_ = false;
  std::vector<std::vector<double>> slab_;
};

class LinearizorPowerVarprojHipMulti : public Linearizor, public StateMirror {
 public:
  LinearizorPowerVarprojHipMulti(BalProblem& bal_problem, const SolverOptions& options, SolverSummary* summary, bool homogeneous)
      : options_(options), bal_problem_(bal_problem), summary_(summary), homogeneous_(homogeneous), team_(options.gpus) {
    using ST = SolverOptions::SolverType;
    if (options.solver_type_step_1 == ST::PCG || options.solver_type_step_1 == ST::CHOLESKY ||
        options.solver_type_step_2 == SolverOptions::SolverTypeRiemannian::RIPCG) {
      std::fprintf(stderr, "FATAL: --gpus > 1 serves the power-series solvers (POWER_VARPROJ, POWER_SCHUR_COMPLEMENT, RIPOBA)\n");
      std::abort();
    }
    const int world = options.gpus;
    povar_options o{};
    o.robust_norm = (int)options.residual.robust_norm;
    o.huber_parameter = options.residual.huber_parameter;
    o.jacobi_scaling_eps = options.jacobi_scaling_epsilon > 0 ? options.jacobi_scaling_epsilon : 1e-5;
    o.e0_mode = options.e0_mode == "tiles" ? POVAR_E0_TILES
                : options.e0_mode == "implicit" ? POVAR_E0_IMPLICIT : POVAR_E0_IMPLICIT_LDSACC;
    o.flags = options.deterministic ? POVAR_FLAG_DETERMINISTIC : 0u;
    std::vector<int> lm_off, cam_idx;
    std::vector<double> obs;
    bal_problem.flatten(lm_off, cam_idx, obs);
    const int n_lms = bal_problem.num_landmarks();
    const int n_dev = std::max(povar_device_count(), 1);
    const bool distinct = world <= n_dev;
    const char* force_host = std::getenv("POVAR_HOST_COMM");
    use_rccl_ = distinct && !(force_host && force_host[0] == '1');
    uint8_t uid[128] = {0};
    if (use_rccl_) check(povar_comm_unique_id(uid), "povar_comm_unique_id");
    std::fprintf(stderr, "[bal] shard team of %d on %d device(s): exchange: %s\n", world, n_dev, use_rccl_ ? "RCCL" : "host all-reduce");
    shards_.resize(world);
    team_.hook_args.resize(world);
    for (int r = 0; r < world; ++r) team_.hook_args[r] = {&team_, r};
    team_.run([&](int r) {
      Shard& s = shards_[r];
      check(povar_shard_range(n_lms, lm_off.data(), world, r, &s.lb, &s.le), "povar_shard_range");
      std::vector<int> off(lm_off.begin() + s.lb, lm_off.begin() + s.le + 1);
      const int ob = off.front();
      for (int& v : off) v -= ob;
      povar_options or_ = o;
      or_.device = (options.device + r) % n_dev;
      check(povar_create(&s.ctx, bal_problem.num_cameras(), s.le - s.lb, (int64_t)off.back(), off.data(), cam_idx.data() + ob,
                         obs.data() + 2 * (size_t)ob, &or_), "povar_create");
      if (use_rccl_) check(povar_comm_init(s.ctx, world, r, uid), "povar_comm_init");
      else check(povar_comm_init_host(s.ctx, world, r, &ShardTeam::allreduce_hook, &team_.hook_args[r]), "povar_comm_init_host");
      check(povar_set_jl_col_scaling(s.ctx, 1), "povar_set_jl_col_scaling");
      if (r == 0) check(povar_timings_enable(s.ctx, 1), "povar_timings_enable");
    });
    push_state();
    bal_problem_.mirror = this;
  }
  ~LinearizorPowerVarprojHipMulti() override {
    pull_state();
    bal_problem_.mirror = nullptr;
    team_.run([&](int r) { povar_destroy(shards_[r].ctx); });
  }
  bool uses_rccl() const { return use_rccl_; }

  void start_iteration(IterationSummary* it) override { it_summary_ = it; }
  void finish_iteration() override {}
  void initialize_varproj_lm_pOSE(double alpha, bool initialization_varproj) override {
    if (initialization_varproj) all([&](Shard& s, int) { check(povar_init_landmarks_pose(s.ctx, alpha), "povar_init_landmarks_pose"); });
  }
  void compute_error_pOSE(ResidualInfo& ri, bool) override {
    std::vector<povar_residual_info> r(shards_.size());
    all([&](Shard& s, int k) { check(povar_error_pose(s.ctx, options_.alpha, &r[k]), "povar_error_pose"); });
    ri = to_ri(r[0]);  // (all-reduced inside the library: the same on every rank)
    IF_SET(it_summary_)->residual_evaluation_time_in_seconds += device_seconds(4);
    IF_SET(summary_)->num_residual_evaluations += 1;
  }
  void compute_error_homogeneous(ResidualInfo& ri, bool) override {
    std::vector<povar_residual_info> r(shards_.size());
    all([&](Shard& s, int k) { check(povar_error_homogeneous(s.ctx, &r[k]), "povar_error_homogeneous"); });
    ri = to_ri(r[0]);
    IF_SET(it_summary_)->residual_evaluation_time_in_seconds += device_seconds(4);
    IF_SET(summary_)->num_residual_evaluations += 1;
  }
  void linearize_pOSE(double alpha) override {
    std::atomic<int> failed{0};
    all([&](Shard& s, int) {
      const int rc = povar_linearize_pose(s.ctx, alpha);
      check(rc, "povar_linearize_pose");
      if (rc == POVAR_NUMERIC_FAILURE) failed.store(1);
    });
    after_linearize(failed.load());
  }
  void linearize_projective_space_homogeneous() override {
    std::atomic<int> failed{0};
    all([&](Shard& s, int) {
      const int rc = povar_linearize_homogeneous(s.ctx);
      check(rc, "povar_linearize_homogeneous");
      if (rc == POVAR_NUMERIC_FAILURE) failed.store(1);
    });
    after_linearize(failed.load());
  }
  VecX solve(const SolverOptions& so, double lambda, double) override {
    const size_t n = 12 * (size_t)bal_problem_.num_cameras();
    std::vector<VecX> inc(shards_.size(), VecX(n));
    std::vector<int32_t> iters(shards_.size(), 0), term(shards_.size(), 0);
    const int st = so.solver_type_step_1 == SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT
                       ? POVAR_POWER_SCHUR_COMPLEMENT : POVAR_POWER_VARPROJ;
    all([&](Shard& s, int k) {
      check(povar_prepare_pose(s.ctx, lambda, st), "povar_prepare_pose");
      check(povar_power_series_pose(s.ctx, options_.power_sc_iterations, options_.eta, options_.r_tolerance, &iters[k], &term[k]),
            "povar_power_series_pose");
      check(povar_get_increment(s.ctx, inc[k].data()), "povar_get_increment");
    });
    IF_SET(it_summary_)->prepare_time_in_seconds = device_seconds(1);
    fill_solver_summary(device_seconds(2), iters[0], term[0]);
    return std::move(inc[0]);
  }
  VecX solve_joint(double lambda, double) override {
    const size_t n = 11 * (size_t)bal_problem_.num_cameras();
    std::vector<VecX> inc(shards_.size(), VecX(n));
    std::vector<int32_t> iters(shards_.size(), 0), term(shards_.size(), 0);
    all([&](Shard& s, int k) {
      check(povar_solve_joint(s.ctx, lambda, options_.power_sc_iterations, options_.eta, options_.r_tolerance, inc[k].data(),
                              &iters[k], &term[k]), "povar_solve_joint");
    });
    IF_SET(it_summary_)->prepare_time_in_seconds = device_seconds(1);
    fill_solver_summary(device_seconds(2), iters[0], term[0]);
    return std::move(inc[0]);
  }
  double apply(const SolverOptions& so, double alpha, VecX&& inc) override {
    std::vector<double> l_diff(shards_.size(), 0.0);
    const int st = so.solver_type_step_1 == SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT
                       ? POVAR_POWER_SCHUR_COMPLEMENT : POVAR_POWER_VARPROJ;
    all([&](Shard& s, int k) { check(povar_apply_pose(s.ctx, st, alpha, inc.data(), &l_diff[k]), "povar_apply_pose"); });
    IF_SET(it_summary_)->back_substitution_time_in_seconds = device_seconds(3);
    return l_diff[0];
  }
  double apply_joint(VecX&& inc) override {
    std::vector<double> l_diff(shards_.size(), 0.0);
    all([&](Shard& s, int k) { check(povar_apply_joint(s.ctx, inc.data(), &l_diff[k]), "povar_apply_joint"); });
    IF_SET(it_summary_)->back_substitution_time_in_seconds = device_seconds(3);
    return l_diff[0];
  }

  // StateMirror
  void backup_pOSE() override { all([&](Shard& s, int) { check(povar_backup_pose(s.ctx), "povar_backup_pose"); }); }
  void restore_pOSE() override { all([&](Shard& s, int) { check(povar_restore_pose(s.ctx), "povar_restore_pose"); }); }
  void backup_joint() override { all([&](Shard& s, int) { check(povar_backup_joint(s.ctx), "povar_backup_joint"); }); }
  void restore_joint() override { all([&](Shard& s, int) { check(povar_restore_joint(s.ctx), "povar_restore_joint"); }); }
  void normalize_joint() override { all([&](Shard& s, int) { check(povar_normalize_joint(s.ctx), "povar_normalize_joint"); }); }
  void pull_state() override {
    const int nc = bal_problem_.num_cameras();
    const int w = homogeneous_ ? 4 : 3;
    std::vector<double> cams(12 * (size_t)nc);
    all([&](Shard& s, int k) {
      if (k == 0) check(povar_get_cameras(s.ctx, cams.data()), "povar_get_cameras");
      std::vector<double> lms((size_t)w * (s.le - s.lb));
      if (homogeneous_) check(povar_get_landmarks_homogeneous(s.ctx, lms.data()), "povar_get_landmarks_homogeneous");
      else check(povar_get_landmarks(s.ctx, lms.data()), "povar_get_landmarks");
      for (int l = s.lb; l < s.le; ++l)
        for (int q = 0; q < w; ++q) {
          const double v = lms[(size_t)w * (l - s.lb) + q];
          if (homogeneous_) bal_problem_.landmarks()[l].p_w_homogeneous[q] = v;
          else bal_problem_.landmarks()[l].p_w[q] = v;
        }
    });
    for (int c = 0; c < nc; ++c)
      for (int q = 0; q < 12; ++q) bal_problem_.cameras()[c].space_matrix[q] = cams[12 * (size_t)c + q];
  }

 private:
  struct Shard {
    povar_ctx* ctx = nullptr;
    int32_t lb = 0, le = 0;
  };
  template <class F>
  void all(F&& f) {
    team_.run([&](int r) { f(shards_[r], r); });
  }
  void after_linearize(int failed) {
    if (failed) {
      std::fprintf(stderr, "FATAL: did not expect numerical failure during linearization\n");
      std::abort();
    }
    const double e = device_seconds(0);
    IF_SET(it_summary_)->jacobian_evaluation_time_in_seconds = e;
    IF_SET(it_summary_)->stage1_time_in_seconds = e;
    IF_SET(summary_)->num_jacobian_evaluations += 1;
  }
  double device_seconds(int kind) {  // rank 0's device time (the ranks run the same calls side by side)
    povar_timings_info t;
    check(povar_timings(shards_[0].ctx, &t), "povar_timings");
    const double now[5] = {t.linearize_ms, t.prepare_ms, t.solve_ms, t.apply_ms, t.other_ms};
    const double d = (now[kind] - seen_[kind]) * 1e-3;
    seen_[kind] = now[kind];
    return d;
  }
  void push_state() {
    const int nc = bal_problem_.num_cameras();
    const int w = homogeneous_ ? 4 : 3;
    std::vector<double> cams(12 * (size_t)nc);
    for (int c = 0; c < nc; ++c)
      for (int q = 0; q < 12; ++q) cams[12 * (size_t)c + q] = bal_problem_.cameras()[c].space_matrix[q];
    all([&](Shard& s, int) {
      check(povar_set_cameras(s.ctx, cams.data()), "povar_set_cameras");
      std::vector<double> lms((size_t)w * (s.le - s.lb));
      for (int l = s.lb; l < s.le; ++l)
        for (int q = 0; q < w; ++q)
          lms[(size_t)w * (l - s.lb) + q] = homogeneous_ ? bal_problem_.landmarks()[l].p_w_homogeneous[q] : bal_problem_.landmarks()[l].p_w[q];
      if (homogeneous_) check(povar_set_landmarks_homogeneous(s.ctx, lms.data()), "povar_set_landmarks_homogeneous");
      else check(povar_set_landmarks(s.ctx, lms.data()), "povar_set_landmarks");
    });
  }
  void fill_solver_summary(double seconds, int iters, int term) {
    IF_SET(it_summary_)->solve_reduced_system_time_in_seconds = seconds;
    IF_SET(it_summary_)->linear_solver_iterations = iters;
    IF_SET(it_summary_)->linear_solver_message =
        term == POVAR_LINEAR_SOLVER_SUCCESS ? "Iteration: " + std::to_string(iters) + " Convergence."
                                            : "Maximum number of iterations reached.";
    IF_SET(it_summary_)->linear_solver_type = "bal_power_sc";
    IF_SET(summary_)->num_linear_solves += 1;
  }

  SolverOptions options_;
  BalProblem& bal_problem_;
  SolverSummary* summary_ = nullptr;
  IterationSummary* it_summary_ = nullptr;
  bool homogeneous_;
  bool use_rccl_ = false;
  ShardTeam team_;
  std::vector<Shard> shards_;
  double seen_[5] = {0, 0, 0, 0, 0};
};

LinearizorFactory g_factory = nullptr;

std::unique_ptr<Linearizor> make(BalProblem& p, const SolverOptions& o, SolverSummary* s, bool hom) {
  if (g_factory) return g_factory(p, o, s, hom);
  // POVAR_FORCE_MULTI=1: also `--gpus 1` goes through the N-shard class (a team of one: an RCCL communicator of one rank
  // created on the worker thread, the all-reduces of the one shard inside its captured term loop with POVAR_GRAPH_COMM=1)
  // -- the branch a one-GPU box can otherwise never execute (tests/test_gpu_bal_cli.py)
  const char* force_multi = std::getenv("POVAR_FORCE_MULTI");
  if (o.gpus > 1 || (force_multi && force_multi[0] == '1')) return std::make_unique<LinearizorPowerVarprojHipMulti>(p, o, s, hom);
  return std::make_unique<LinearizorPowerVarprojHip>(p, o, s, hom);
}

}  // namespace

void set_linearizor_factory(LinearizorFactory f) { g_factory = f; }

std::unique_ptr<Linearizor> Linearizor::create(BalProblem& p, const SolverOptions& o, SolverSummary* s) {
  return make(p, o, s, false);  // solver/linearizor.cpp:47-62: every SolverType has a device implementation
}

std::unique_ptr<Linearizor> Linearizor::create_homogeneous(BalProblem& p, const SolverOptions& o, SolverSummary* s) {
  return make(p, o, s, true);  // solver/linearizor.cpp:64-80
}

}  // namespace povar_host
