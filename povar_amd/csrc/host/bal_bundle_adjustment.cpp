// LM / VarPro outer loops of the drop-in surface, restated from
// solver/bal_bundle_adjustment.cpp: optimize_lm_ours_pOSE (252-542), create_homogeneous_landmark
// (545-553), optimize_homogeneous_joint (557-843), bundle_adjust_manual (848-876).  The two loops of
// the reference differ only at the points marked [step 1] / [step 2]; they share one body here.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>

#include "linearizor.hpp"

namespace povar_host {

namespace {

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  void reset() { t0 = std::chrono::steady_clock::now(); }
};

std::string item_oneline(const ResidualItem& it) {  // residual_info.cpp:77-80
  char buf[128];
  std::snprintf(buf, sizeof buf, "%.4e (mean res: %.2f, num: %ld)", it.error, it.residual_mean(), it.num_obs);
  return buf;
}

// finish_iteration, bal_bundle_adjustment.cpp:61-94
void finish_iteration(SolverSummary& summary, IterationSummary& it) {
  it.step_solver_time_in_seconds = it.scale_landmark_jacobian_time_in_seconds + it.stage2_time_in_seconds +
                                   it.solve_reduced_system_time_in_seconds + it.back_substitution_time_in_seconds;
  if (it.iteration > 0 && !summary.iterations.empty())
    it.cost_change_all_error = summary.iterations.back().cost.all.error - it.cost.all.error;
  summary.iterations.push_back(it);
  std::fflush(stdout);
}

// finish_solve, bal_bundle_adjustment.cpp:97-159 (switches on solver_type_step_1 only: quirk A.6)
void finish_solve(SolverSummary& summary, const SolverOptions& options) {
  switch (options.solver_type_step_1) {
    case SolverOptions::SolverType::PCG: summary.solver_type = "bal_pcg"; break;
    case SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT: summary.solver_type = "bal_power_sc"; break;
    case SolverOptions::SolverType::POWER_VARPROJ: summary.solver_type = "power_variable_projection"; break;
    default: summary.solver_type = "variable_projection";
  }
  summary.initial_cost = summary.iterations.front().cost;
  for (auto it = summary.iterations.rbegin(); it != summary.iterations.rend(); ++it)
    if (it->step_is_successful) { summary.final_cost = it->cost; break; }
  summary.num_successful_steps = -1;
  summary.num_unsuccessful_steps = 0;
  summary.linear_solver_time_in_seconds = summary.residual_evaluation_time_in_seconds =
      summary.jacobian_evaluation_time_in_seconds = 0;
  for (const auto& it : summary.iterations) {
    if (it.step_is_successful) ++summary.num_successful_steps; else ++summary.num_unsuccessful_steps;
    summary.linear_solver_time_in_seconds += it.step_solver_time_in_seconds;
    summary.residual_evaluation_time_in_seconds += it.residual_evaluation_time_in_seconds;
    summary.jacobian_evaluation_time_in_seconds += it.jacobian_evaluation_time_in_seconds;
  }
}

// compute_cost_decrease, bal_bundle_adjustment.cpp:163-176
double compute_cost_decrease(const ResidualInfo& a, const ResidualInfo& b, SolverOptions::OptimizedCost oc) {
  switch (oc) {
    case SolverOptions::OptimizedCost::ERROR: return a.all.error - b.all.error;
    case SolverOptions::OptimizedCost::ERROR_VALID: return a.valid.error - b.valid.error;
    default: return a.valid.error_avg() - b.valid.error_avg();
  }
}

// function_tolerance_reached, bal_bundle_adjustment.cpp:179-205
bool function_tolerance_reached(const IterationSummary& it, const SolverOptions& o, std::string& message) {
  // cost_change is only tracked for `all` here; ERROR_VALID* coincide with it on pOSE (validity always true)
  const double cost = o.optimized_cost == SolverOptions::OptimizedCost::ERROR ? it.cost.all.error : it.cost.valid.error;
  const double change = std::abs(it.cost_change_all_error);
  if (change <= o.function_tolerance * cost) {
    char buf[160];
    std::snprintf(buf, sizeof buf, "Function tolerance reached. |cost_change|/cost: %g <= %g", change / cost, o.function_tolerance);
    message = buf;
    return true;
  }
  return false;
}

// format_new_error_info, bal_bundle_adjustment.cpp:208-226
std::string format_new_error_info(const ResidualInfo& ri, SolverOptions::OptimizedCost oc) {
  char buf[160];
  switch (oc) {
    case SolverOptions::OptimizedCost::ERROR:
      std::snprintf(buf, sizeof buf, "error: %.4e (mean res: %.2f, num valid: %ld)", ri.all.error, ri.all.residual_mean(), ri.valid.num_obs);
      break;
    case SolverOptions::OptimizedCost::ERROR_VALID:
      std::snprintf(buf, sizeof buf, "error valid: %.4e (mean res: %.2f, num: %ld)", ri.valid.error, ri.valid.residual_mean(), ri.valid.num_obs);
      break;
    default:
      std::snprintf(buf, sizeof buf, "error valid avg: %.4e (mean res: %.2f, num: %ld)", ri.valid.error_avg(), ri.valid.residual_mean(), ri.valid.num_obs);
  }
  return buf;
}

// check_options, bal_bundle_adjustment.cpp:228-250
void check_options(const SolverOptions& o) {
  if (!(o.min_trust_region_radius <= o.initial_trust_region_radius && o.initial_trust_region_radius <= o.max_trust_region_radius) ||
      o.jacobi_scaling_epsilon < 0) {
    std::fprintf(stderr, "FATAL: Invalid configuration\n");
    std::abort();
  }
}

void optimize_lm(BalProblem& bal_problem, const SolverOptions& so, SolverSummary& summary, const Timer& timer_total,
                 bool step2) {
  Timer timer_preprocessor;
  const double min_lambda = 1.0 / so.max_trust_region_radius;
  const double max_lambda = 1.0 / so.min_trust_region_radius;
  const int max_lm_iter = step2 ? so.max_num_iterations_step_2 : so.max_num_iterations_step_1;
  double lambda = 1.0 / so.initial_trust_region_radius;  // step 2 restarts at lambda_0 (quirk A.6)
  double lambda_vee = so.initial_vee;
  check_options(so);
  if (!step2) summary = SolverSummary();  // [step 1] :278; step 2 keeps appending to the same summary (:581-583)
  summary.num_linear_solves = summary.num_residual_evaluations = summary.num_jacobian_evaluations = 0;
  std::unique_ptr<Linearizor> linearizor =
      step2 ? Linearizor::create_homogeneous(bal_problem, so, &summary) : Linearizor::create(bal_problem, so, &summary);
  summary.preprocessor_time_in_seconds = timer_preprocessor.elapsed();
  Timer timer_minimizer;

  bool terminated = false;
  double relative_error_change = 1;
  bool initialization_varproj = !step2;
  auto compute_error = [&](ResidualInfo& ri, bool init) {
    if (step2) linearizor->compute_error_homogeneous(ri, false); else linearizor->compute_error_pOSE(ri, init);
  };
  auto lambda_exceeded = [&]() {
    if (lambda > max_lambda) {
      terminated = true;
      summary.termination_type = NO_CONVERGENCE;
      char buf[128];
      std::snprintf(buf, sizeof buf, "Solver did not converge and reached maximum damping lambda of %g", max_lambda);
      summary.message = buf;
    }
  };

  for (int it = 0; it <= max_lm_iter && !terminated;) {
    IterationSummary it_summary;
    it_summary.iteration = it;
    linearizor->start_iteration(&it_summary);
    Timer timer_iteration;
    ResidualInfo ri;
    if (initialization_varproj) linearizor->initialize_varproj_lm_pOSE(so.alpha, initialization_varproj);  // [step 1] :302-304
    compute_error(ri, initialization_varproj);
    initialization_varproj = false;
    std::printf("Iteration %d, %s\n", it, error_summary_oneline(ri, so.use_projection_validity_check()).c_str());
    if (!ri.is_numerically_valid) {
      std::fprintf(stderr, "FATAL: did not expect numerical failure during linearization\n");
      std::abort();
    }
    if (it == 0) {  // iteration 0 is just error evaluation and logging, :316-328
      linearizor->finish_iteration();
      it_summary.cost = ri;
      it_summary.trust_region_radius = 1 / lambda;
      it_summary.iteration_time_in_seconds = timer_iteration.elapsed();
      it_summary.cumulative_time_in_seconds = timer_total.elapsed();
      it_summary.step_is_successful = true;
      it_summary.step_is_valid = true;
      finish_iteration(summary, it_summary);
      ++it;
      continue;
    }
    if (step2) linearizor->linearize_projective_space_homogeneous(); else linearizor->linearize_pOSE(so.alpha);
    std::printf("\t[INFO] Stage 1 time %.3fs.\n", it_summary.stage1_time_in_seconds);

    for (int j = 0; j < std::numeric_limits<int>::max() && it <= max_lm_iter && !terminated; j++) {
      if (j > 0) {
        std::printf("Iteration %d, backtracking\n", it);
        it_summary = IterationSummary();
        it_summary.iteration = it;
        linearizor->start_iteration(&it_summary);
        timer_iteration.reset();
      }
      Linearizor::VecX inc = step2 ? linearizor->solve_joint(lambda, relative_error_change)
                                   : linearizor->solve(so, lambda, relative_error_change);
      std::printf("\t[INFO] Stage 2 time %.3fs.\n", it_summary.stage2_time_in_seconds);
      std::printf("\t[CG] Summary: %s Time %.3fs. Time per iteration %.6fs\n", it_summary.linear_solver_message.c_str(),
                  it_summary.solve_reduced_system_time_in_seconds,
                  it_summary.solve_reduced_system_time_in_seconds / std::max(it_summary.linear_solver_iterations, 1));

      const bool finite = std::all_of(inc.begin(), inc.end(), [](double v) { return std::isfinite(v); });
      if (!finite) {  // :362-401
        it_summary.step_is_valid = false;
        it_summary.step_is_successful = false;
        const double iteration_time = timer_iteration.elapsed(), cumulative_time = timer_total.elapsed();
        std::printf("\t[Invalid] Numeric issues when computing increment (contains NaNs), lambda: %.1e, cg_iter: %d, it_time: %.3fs, total_time: %.3fs\n",
                    lambda, it_summary.linear_solver_iterations, iteration_time, cumulative_time);
        lambda = lambda_vee * lambda;
        lambda_vee *= so.vee_factor;
        linearizor->finish_iteration();
        it_summary.trust_region_radius = 1 / lambda;
        it_summary.iteration_time_in_seconds = iteration_time;
        it_summary.cumulative_time_in_seconds = cumulative_time;
        finish_iteration(summary, it_summary);
        it++;
        lambda_exceeded();
        continue;
      }
      if (step2) bal_problem.backup_joint(); else bal_problem.backup_pOSE();
      double l_diff = step2 ? linearizor->apply_joint(std::move(inc)) : linearizor->apply(so, so.alpha, std::move(inc));
      if (step2 && bal_problem.mirror) bal_problem.mirror->normalize_joint();  // [step 2] :700-705
      else if (step2) {
        for (auto& c : bal_problem.cameras()) {
          double s = 0;
          for (double v : c.space_matrix) s += v * v;
          s = std::sqrt(s);
          for (double& v : c.space_matrix) v /= s;
        }
        for (auto& l : bal_problem.landmarks()) {
          const double w = l.p_w_homogeneous[3];
          for (double& v : l.p_w_homogeneous) v /= w;
        }
      }
      ResidualInfo ri2;
      compute_error(ri2, false);
      it_summary.cost = ri2;
      relative_error_change = std::abs(ri.all.error - ri2.all.error) / ri.all.error;
      if (!ri2.is_numerically_valid) {
        it_summary.step_is_valid = false;
        it_summary.step_is_successful = false;
        std::printf("\t[EVAL] failed to evaluate cost: %s", error_summary_oneline(ri2, so.use_projection_validity_check()).c_str());
      } else {
        const double f_diff = compute_cost_decrease(ri, ri2, so.optimized_cost);
        if (so.optimized_cost == SolverOptions::OptimizedCost::ERROR_VALID_AVG) l_diff /= ri.valid.num_obs;
        const double step_quality = f_diff / l_diff;
        std::printf("\t[EVAL] f_diff %.4e ri1 %.4e ri2 %.4e\n", f_diff, ri.valid.error, ri2.valid.error);
        it_summary.relative_decrease = step_quality;
        if (step2) {  // [step 2] :742-745
          it_summary.step_is_valid = l_diff > 0;
          it_summary.step_is_successful = it_summary.step_is_valid && step_quality > so.min_relative_decrease;
        } else {      // [step 1] :442-445 -- only f_diff > 0 is required
          it_summary.step_is_valid = true;
          it_summary.step_is_successful = f_diff > 0;
        }
      }
      const double iteration_time = timer_iteration.elapsed(), cumulative_time = timer_total.elapsed();
      if (it_summary.step_is_successful) {
        std::printf("\t[Success] %s, lambda: %.1e, cg_iter: %d, it_time: %.3fs, total_time: %.3fs\n",
                    format_new_error_info(ri2, so.optimized_cost).c_str(), lambda, it_summary.linear_solver_iterations,
                    iteration_time, cumulative_time);
        lambda *= std::max(1.0 / 3, 1 - std::pow(2 * it_summary.relative_decrease - 1, 3));  // :461-462
        lambda = std::max(min_lambda, lambda);
        lambda_vee = so.initial_vee;
        linearizor->finish_iteration();
        it_summary.trust_region_radius = 1 / lambda;
        it_summary.iteration_time_in_seconds = iteration_time;
        it_summary.cumulative_time_in_seconds = cumulative_time;
        finish_iteration(summary, it_summary);
        it++;
        if (function_tolerance_reached(summary.iterations.back(), so, summary.message)) {
          terminated = true;
          summary.termination_type = CONVERGENCE;
        }
        break;
      } else {
        std::printf("\t[%s] %s, lambda: %.1e, cg_iter: %d, it_time: %.3fs, total_time: %.3fs\n",
                    it_summary.step_is_valid ? "Reject" : "Invalid", format_new_error_info(ri2, so.optimized_cost).c_str(),
                    lambda, it_summary.linear_solver_iterations, iteration_time, cumulative_time);
        lambda = lambda_vee * lambda;
        lambda_vee *= so.vee_factor;
        linearizor->finish_iteration();
        it_summary.trust_region_radius = 1 / lambda;
        it_summary.iteration_time_in_seconds = iteration_time;
        it_summary.cumulative_time_in_seconds = cumulative_time;
        it_summary.step_is_successful = false;
        finish_iteration(summary, it_summary);
        if (step2) bal_problem.restore_joint(); else bal_problem.restore_pOSE();
        it++;
        lambda_exceeded();
      }
    }
  }
  if (!terminated) {
    summary.termination_type = NO_CONVERGENCE;
    char buf[128];
    std::snprintf(buf, sizeof buf, "Solver did not converge after maximum number of %d iterations", max_lm_iter);
    summary.message = buf;
  }
  summary.minimizer_time_in_seconds = timer_minimizer.elapsed();
  summary.total_time_in_seconds = timer_total.elapsed();
  finish_solve(summary, so);
  std::printf("Final Cost: %s\n", error_summary_oneline(summary.final_cost, so.use_projection_validity_check()).c_str());
  std::printf("%s: %s\n", summary.termination_type == CONVERGENCE ? "CONVERGENCE" : "NO_CONVERGENCE", summary.message.c_str());
  std::fflush(stdout);
  // the linearizor (and its device context) is destroyed here: state flows back into bal_problem
}

// create_homogeneous_landmark, bal_bundle_adjustment.cpp:545-553
void create_homogeneous_landmark(BalProblem& bal_problem) {
  for (auto& l : bal_problem.landmarks()) l.p_w_homogeneous = {l.p_w[0], l.p_w[1], l.p_w[2], 1.0};
  for (auto& c : bal_problem.cameras()) {
    double s = 0;
    for (double v : c.space_matrix) s += v * v;
    s = std::sqrt(s);
    for (double& v : c.space_matrix) v /= s;
  }
}

}  // namespace

std::string error_summary_oneline(const ResidualInfo& info, bool valid_first) {  // residual_info.cpp:82-94
  const std::string warning = info.is_numerically_valid ? "" : "!NaN! ";
  if (valid_first) return warning + "error valid: " + item_oneline(info.valid) + ", error: " + item_oneline(info.all);
  return warning + "error: " + item_oneline(info.all) + ", error valid: " + item_oneline(info.valid);
}

void bundle_adjust_manual(BalProblem& bal_problem, const SolverOptions& so, SolverSummary* out) {
  SolverSummary local;
  SolverSummary& summary = out ? *out : local;
  Timer timer_total;
  optimize_lm(bal_problem, so, summary, timer_total, false);   // first step: linear VarPro, :860
  create_homogeneous_landmark(bal_problem);                    // :861
  optimize_lm(bal_problem, so, summary, timer_total, true);    // second step: Riemannian manifold optimisation, :864
}

// ba_log.json: the per-iteration arrays of the reference's log (bal/ba_log.hpp:147-245,
// ba_log.cpp:72-114), flat arrays + _static + _type
void save_ba_log_json(const SolverSummary& s, const SolverOptions& o, const BalProblem& p) {
  if (o.log.disable_all || o.log.log_path.empty()) return;
  FILE* f = std::fopen(o.log.log_path.c_str(), "w");
  if (!f) return;
  auto arr_d = [&](const char* name, auto get, bool last = false) {
    std::fprintf(f, "  \"%s\": [", name);
    for (size_t i = 0; i < s.iterations.size(); ++i) std::fprintf(f, "%s%.17g", i ? ", " : "", (double)get(s.iterations[i]));
    std::fprintf(f, "]%s\n", last ? "" : ",");
  };
  std::fprintf(f, "{\n  \"_type\": \"rootba_povar\",\n");
  std::fprintf(f, "  \"_static\": {\"problem_info\": {\"num_cameras\": %d, \"num_landmarks\": %d, \"num_observations\": %ld},"
                  " \"solver\": {\"solver_type\": \"%s\", \"termination_type\": \"%s\", \"message\": \"%s\","
                  " \"num_successful_steps\": %d, \"num_unsuccessful_steps\": %d, \"num_linear_solves\": %d,"
                  " \"total_time_in_seconds\": %.6f}},\n",
               p.num_cameras(), p.num_landmarks(), p.num_observations(), s.solver_type.c_str(),
               s.termination_type == CONVERGENCE ? "CONVERGENCE" : "NO_CONVERGENCE", s.message.c_str(),
               s.num_successful_steps, s.num_unsuccessful_steps, s.num_linear_solves, s.total_time_in_seconds);
  arr_d("iteration", [](const IterationSummary& i) { return i.iteration; });
  arr_d("cost", [](const IterationSummary& i) { return i.cost.all.error; });
  arr_d("cost_valid", [](const IterationSummary& i) { return i.cost.valid.error; });
  arr_d("step_is_successful", [](const IterationSummary& i) { return i.step_is_successful ? 1 : 0; });
  arr_d("step_is_valid", [](const IterationSummary& i) { return i.step_is_valid ? 1 : 0; });
  arr_d("relative_decrease", [](const IterationSummary& i) { return i.relative_decrease; });
  arr_d("trust_region_radius", [](const IterationSummary& i) { return i.trust_region_radius; });
  arr_d("linear_solver_iterations", [](const IterationSummary& i) { return i.linear_solver_iterations; });
  arr_d("cumulative_time", [](const IterationSummary& i) { return i.cumulative_time_in_seconds; });
  arr_d("iteration_time", [](const IterationSummary& i) { return i.iteration_time_in_seconds; });
  arr_d("stage1_time", [](const IterationSummary& i) { return i.stage1_time_in_seconds; });
  arr_d("prepare_time", [](const IterationSummary& i) { return i.prepare_time_in_seconds; });
  arr_d("solve_reduced_system_time", [](const IterationSummary& i) { return i.solve_reduced_system_time_in_seconds; });
  arr_d("back_substitution_time", [](const IterationSummary& i) { return i.back_substitution_time_in_seconds; }, true);
  std::fprintf(f, "}\n");
  std::fclose(f);
}

}  // namespace povar_host
