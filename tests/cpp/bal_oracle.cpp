// TEST INFRASTRUCTURE: the restated `bal` host program (povar_amd/csrc/host) driven by a Linearizor
// backed by the CPU oracle (oracle/povar_oracle.c) instead of the HIP library.  Same LM loop, same
// loader, same CLI: the end-to-end test runs this binary and bin/bal on the same input file and
// compares the iteration logs.  Mirrors LinearizorPowerVarproj (solver/linearizor_power_varproj.cpp) and,
// for --solver-type-step-1 PCG | CHOLESKY / --solver-type-step-2 RIPCG, LinearizorSC (solver/linearizor_sc.cpp).
#include <cmath>
#include <cstdio>

#include "../../oracle/povar_oracle.h"
#include "../../povar_amd/csrc/host/linearizor.hpp"

using namespace povar_host;

namespace {

class LinearizorOracle : public Linearizor {
 public:
  LinearizorOracle(BalProblem& p, const SolverOptions& o, SolverSummary* s, bool hom)
      : options_(o), bal_(p), summary_(s), hom_(hom) {
    p.flatten(lm_off_, cam_idx_, obs_);
    prob_ = {p.num_cameras(), p.num_landmarks(), (int64_t)cam_idx_.size(), lm_off_.data(), cam_idx_.data(), obs_.data()};
    opts_ = {(int)o.residual.robust_norm, o.residual.huber_parameter,
             o.jacobi_scaling_epsilon > 0 ? o.jacobi_scaling_epsilon : 1e-5};
    nc_ = p.num_cameras();
    nl_ = p.num_landmarks();
    storage_.resize(64 * cam_idx_.size());
    storage_h_.resize(34 * cam_idx_.size());
    storage_n_.resize(28 * cam_idx_.size());
    hll_.resize(9 * (size_t)nl_);
    binv_.resize(144 * (size_t)nc_);
    jls_.resize(4 * (size_t)nl_);
    sigma_.resize(12 * (size_t)nc_);
    using ST = SolverOptions::SolverType;
    sc1_ = !hom && (o.solver_type_step_1 == ST::PCG || o.solver_type_step_1 == ST::CHOLESKY);
    sc2_ = hom && o.solver_type_step_2 == SolverOptions::SolverTypeRiemannian::RIPCG;
  }
  void start_iteration(IterationSummary* it) override { it_ = it; }
  void finish_iteration() override {}
  void initialize_varproj_lm_pOSE(double alpha, bool init) override {
    if (!init) return;
    pull();
    orc_init_landmarks_pose(&prob_, alpha, cams_.data(), lms_.data());
    push();
  }
  void compute_error_pOSE(ResidualInfo& ri, bool) override {
    pull();
    orc_residual_info r;
    orc_error_pose(&prob_, &opts_, options_.alpha, cams_.data(), lms_.data(), &r);
    ri = conv(r);
  }
  void compute_error_homogeneous(ResidualInfo& ri, bool) override {
    pull();
    orc_residual_info r;
    orc_error_homogeneous(&prob_, &opts_, cams_.data(), lms_.data(), &r);
    ri = conv(r);
  }
  void linearize_pOSE(double alpha) override {  // cpp:45-76
    pull();
    orc_linearize_pose(&prob_, &opts_, alpha, cams_.data(), lms_.data(), storage_.data());
    std::vector<double> d2(12 * (size_t)nc_);
    orc_jp_diag2_pose(&prob_, storage_.data(), d2.data());
    if (!sc1_) orc_scale_jl_cols_pose(&prob_, &opts_, storage_.data(), jls_.data());  // linearizor_sc.cpp:163-191 skips it
    for (size_t i = 0; i < d2.size(); ++i) sigma_[i] = 1.0 / (opts_.jacobi_scaling_eps + std::sqrt(d2[i]));
    new_lin_ = true;
  }
  void linearize_projective_space_homogeneous() override {  // cpp:80-110
    pull();
    orc_linearize_homogeneous(&prob_, &opts_, cams_.data(), lms_.data(), storage_h_.data());
    std::vector<double> d2(12 * (size_t)nc_);
    orc_jp_diag2_homogeneous(&prob_, storage_h_.data(), d2.data());
    orc_scale_jl_cols_homogeneous(&prob_, &opts_, storage_h_.data(), jls_.data());
    for (size_t i = 0; i < d2.size(); ++i) sigma_[i] = 1.0 / (opts_.jacobi_scaling_eps + std::sqrt(d2[i]));
    new_lin_ = true;
  }
  VecX solve(const SolverOptions& so, double lambda, double) override {  // cpp:178-243
    if (new_lin_) orc_scale_jp_cols_pose(&prob_, storage_.data(), sigma_.data());
    new_lin_ = false;
    if (sc1_) return solve_sc(12, so.solver_type_step_1 == SolverOptions::SolverType::CHOLESKY, lambda);  // linearizor_sc.cpp:85-160
    const bool poba = so.solver_type_step_1 == SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT;
    std::vector<double> b(12 * (size_t)nc_);
    orc_prepare_hb_pose(&prob_, storage_.data(), lambda, poba ? lambda : 0.0, hll_.data(), b.data(), binv_.data());
    lambda_ = lambda;
    VecX inc(12 * (size_t)nc_);
    int32_t iters = 0;
    const int st = orc_solve_pose(&prob_, storage_.data(), hll_.data(), binv_.data(), b.data(), options_.power_sc_iterations,
                                  options_.eta, options_.r_tolerance, inc.data(), &iters, nullptr, 1);
    report(iters, st);
    return inc;
  }
  VecX solve_joint(double lambda, double) override {  // cpp:114-175
    pull();
    if (new_lin_) {
      orc_scale_jp_cols_joint(&prob_, storage_h_.data(), sigma_.data());
      orc_linearize_nullspace(&prob_, cams_.data(), lms_.data(), storage_h_.data(), storage_n_.data());
    }
    new_lin_ = false;
    if (sc2_) return solve_sc(11, false, lambda);  // linearizor_sc.cpp:224-303
    std::vector<double> b(11 * (size_t)nc_);
    orc_prepare_hb_joint(&prob_, storage_h_.data(), storage_n_.data(), lambda, hll_.data(), b.data(), binv_.data());
    lambda_ = lambda;
    VecX inc(11 * (size_t)nc_);
    int32_t iters = 0;
    const int st = orc_solve_joint(&prob_, storage_n_.data(), hll_.data(), binv_.data(), b.data(), options_.power_sc_iterations,
                                   options_.eta, options_.r_tolerance, inc.data(), &iters, nullptr);
    report(iters, st);
    return inc;
  }
  double apply(const SolverOptions& so, double alpha, VecX&& inc) override {  // cpp:246-273
    pull();
    double l_diff;
    if (so.solver_type_step_1 != SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT) {  // also LinearizorSC::apply, linearizor_sc.cpp:65-83
      for (size_t i = 0; i < inc.size(); ++i) inc[i] *= sigma_[i];
      orc_apply_cam_inc(nc_, cams_.data(), inc.data());
      for (size_t i = 0; i < inc.size(); ++i) inc[i] *= 1.0 / sigma_[i];
      l_diff = orc_back_substitute_pose(&prob_, alpha, storage_.data(), cams_.data(), lms_.data(), inc.data());
    } else {
      l_diff = orc_back_substitute_poba(&prob_, storage_.data(), jls_.data(), lambda_, lms_.data(), inc.data());
      for (size_t i = 0; i < inc.size(); ++i) inc[i] *= sigma_[i];
      orc_apply_cam_inc(nc_, cams_.data(), inc.data());
    }
    push();
    return l_diff;
  }
  double apply_joint(VecX&& inc) override {  // cpp:277-308
    pull();
    const double l_diff = orc_back_substitute_joint(&prob_, storage_h_.data(), jls_.data(), lambda_, cams_.data(), lms_.data(), inc.data());
    orc_apply_cam_inc_joint(nc_, cams_.data(), inc.data(), sigma_.data());
    push();
    return l_diff;
  }

 private:
  VecX solve_sc(int dim, bool direct, double lambda) {
    const size_t n = (size_t)dim * (size_t)nc_;
    std::vector<double> S(n * n), b(n), minv((size_t)dim * dim * nc_);
    if (dim == 12) orc_get_hb_pose(&prob_, storage_.data(), lambda, S.data(), b.data());
    else orc_get_hb_joint(&prob_, storage_h_.data(), storage_n_.data(), lambda, S.data(), b.data());
    lambda_ = lambda;
    VecX inc(n);
    int32_t iters = 0;
    int st = ORC_SUCCESS;
    if (direct) {
      if (orc_cholesky_solve((int32_t)n, S.data(), b.data(), inc.data()))
        for (double& v : inc) v = std::nan("");
    } else {
      orc_block_jacobi_inverse(nc_, dim, S.data(), minv.data());
      st = orc_pcg(nc_, dim, S.data(), b.data(), minv.data(), options_.min_linear_solver_iterations,
                   options_.max_linear_solver_iterations, options_.eta, inc.data(), &iters);
    }
    if (it_) {
      it_->linear_solver_iterations = iters;
      it_->linear_solver_message = direct ? "" : st == ORC_SUCCESS ? "Iteration: " + std::to_string(iters) + " Convergence."
                                   : st == ORC_FAILURE ? "Numerical failure." : "Maximum number of iterations reached.";
      it_->linear_solver_type = "bal_sc";
    }
    if (summary_) summary_->num_linear_solves += 1;
    return inc;
  }
  static ResidualInfo conv(const orc_residual_info& r) {
    ResidualInfo o;
    o.all = {(long)r.all_num_obs, r.all_error, r.all_residual_sum};
    o.valid = {(long)r.valid_num_obs, r.valid_error, r.valid_residual_sum};
    o.is_numerically_valid = r.is_numerically_valid != 0;
    return o;
  }
  void report(int iters, int st) {
    if (it_) {
      it_->linear_solver_iterations = iters;
      it_->linear_solver_message = st == ORC_SUCCESS ? "Iteration: " + std::to_string(iters) + " Convergence."
                                                     : "Maximum number of iterations reached.";
    }
    if (summary_) summary_->num_linear_solves += 1;
  }
  void pull() {
    const int w = hom_ ? 4 : 3;
    cams_.resize(12 * (size_t)nc_);
    lms_.resize(w * (size_t)nl_);
    for (int c = 0; c < nc_; ++c)
      for (int k = 0; k < 12; ++k) cams_[12 * (size_t)c + k] = bal_.cameras()[c].space_matrix[k];
    for (int l = 0; l < nl_; ++l)
      for (int k = 0; k < w; ++k)
        lms_[w * (size_t)l + k] = hom_ ? bal_.landmarks()[l].p_w_homogeneous[k] : bal_.landmarks()[l].p_w[k];
  }
  void push() {
    const int w = hom_ ? 4 : 3;
    for (int c = 0; c < nc_; ++c)
      for (int k = 0; k < 12; ++k) bal_.cameras()[c].space_matrix[k] = cams_[12 * (size_t)c + k];
    for (int l = 0; l < nl_; ++l)
      for (int k = 0; k < w; ++k)
        (hom_ ? bal_.landmarks()[l].p_w_homogeneous[k] : bal_.landmarks()[l].p_w[k]) = lms_[w * (size_t)l + k];
  }

  SolverOptions options_;
  BalProblem& bal_;
  SolverSummary* summary_;
  IterationSummary* it_ = nullptr;
  bool hom_, new_lin_ = false, sc1_ = false, sc2_ = false;
  int nc_ = 0, nl_ = 0;
  double lambda_ = 0;
  std::vector<int> lm_off_, cam_idx_;
  std::vector<double> obs_, cams_, lms_, storage_, storage_h_, storage_n_, hll_, binv_, jls_, sigma_;
  orc_problem prob_{};
  orc_options opts_{};
};

std::unique_ptr<Linearizor> make_oracle(BalProblem& p, const SolverOptions& o, SolverSummary* s, bool hom) {
  return std::make_unique<LinearizorOracle>(p, o, s, hom);
}

}  // namespace

int main(int argc, char** argv) {
  BalAppOptions options;
  if (!parse_bal_app_arguments(argc, argv, options)) return 1;
  set_linearizor_factory(make_oracle);
  // src/app/bal.cpp:83-99: load (+ dataset summary, timing) -> solve -> postprocess -> log
  BalPipelineSummary summary;
  BalProblem bal_problem = load_normalized_bal_problem(options.dataset, &summary.dataset, &summary.timing);
  bundle_adjust_manual(bal_problem, options.solver, &summary.solver, &summary.timing);
  save_ba_log_json(summary, options.solver);
  return 0;
}
