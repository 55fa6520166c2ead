// LinearizorPowerVarproj (solver/linearizor_power_varproj.cpp:21-308 + solver/linearizor_base.cpp:48-100)
// and LinearizorSC (solver/linearizor_sc.cpp:50-360; PCG / CHOLESKY / RIPCG) on top of the C ABI of
// include/povar_hip.h: the two reference classes differ in solve()/solve_joint() and in the Jl column
// scaling of step 1 only, so one class serves both.  The device context owns cameras and landmarks
// between calls; BalProblem forwards backup/restore/normalise through StateMirror.
#include <chrono>
#include <cmath>
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
  std::snprintf(buf, sizeof buf, "%d/%d/%ld/%016llx/norm%d/%.17g/%.17g/dev%d/e0%d", p.num_cameras(), p.num_landmarks(), n_obs, h,
                o.robust_norm, o.huber_parameter, o.jacobi_scaling_eps, o.device, o.e0_mode);
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

LinearizorFactory g_factory = nullptr;

std::unique_ptr<Linearizor> make(BalProblem& p, const SolverOptions& o, SolverSummary* s, bool hom) {
  if (g_factory) return g_factory(p, o, s, hom);
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
