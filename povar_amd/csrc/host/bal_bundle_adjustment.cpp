// LM / VarPro outer loops of the drop-in surface, restated from
// solver/bal_bundle_adjustment.cpp: optimize_lm_ours_pOSE (252-542), create_homogeneous_landmark
// (545-553), optimize_homogeneous_joint (557-843), bundle_adjust_manual (848-876).  The two loops of
// the reference differ only at the points marked [step 1] / [step 2]; they share one body here.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>
#include <fstream>
#include <thread>

#include <sys/resource.h>
#include <unistd.h>

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
  if (it.iteration > 0 && !summary.iterations.empty()) {  // cost.compared_to(previous), residual_info.cpp:43-61
    const ResidualInfo& prev = summary.iterations.back().cost;
    it.cost_change_all_error = prev.all.error - it.cost.all.error;
    it.cost_change_all = {prev.all.num_obs - it.cost.all.num_obs, prev.all.error - it.cost.all.error,
                          prev.all.error_avg() - it.cost.all.error_avg()};
    it.cost_change_valid = {prev.valid.num_obs - it.cost.valid.num_obs, prev.valid.error - it.cost.valid.error,
                            prev.valid.error_avg() - it.cost.valid.error_avg()};
  }
  get_memory_info(it.resident_memory, it.resident_memory_peak);
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
  summary.logging_time_in_seconds = 0;  // "currently this is not computed", :136
  unsigned long long rss = 0;
  get_memory_info(rss, summary.resident_memory_peak);
  // the reference reports TBB's arena concurrency / its observed peak; here one host thread drives the GPU
  summary.num_threads_available = (int)std::thread::hardware_concurrency();
  summary.num_threads_given = options.num_threads;
  summary.num_threads_used = 1;
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

void bundle_adjust_manual(BalProblem& bal_problem, const SolverOptions& so, SolverSummary* out,
                          PipelineTimingSummary* timing) {
  SolverSummary local;
  SolverSummary& summary = out ? *out : local;
  Timer timer_total;
  optimize_lm(bal_problem, so, summary, timer_total, false);   // first step: linear VarPro, :860
  create_homogeneous_landmark(bal_problem);                    // :861
  optimize_lm(bal_problem, so, summary, timer_total, true);    // second step: Riemannian manifold optimisation, :864
  if (timing) timing->optimize_time = summary.total_time_in_seconds;  // :868-870
  // the device context the last linearizor left behind for a successor (step 2 takes over step 1's): nobody comes after
  bal_problem.device_cache.clear();
}

bool get_memory_info(unsigned long long& resident, unsigned long long& resident_peak) {
  std::ifstream fs("/proc/self/statm");
  if (fs.fail()) return false;
  unsigned long long program_size = 0, resident_size = 0;
  fs >> program_size >> resident_size;
  resident = resident_size * (unsigned long long)sysconf(_SC_PAGESIZE);
  struct rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  resident_peak = (unsigned long long)ru.ru_maxrss * 1024;
  return true;
}

void summarize_problem(const BalProblem& p, const std::string& input_path, bool compute_sparsity, DatasetSummary& out) {
  out.type = "bal";
  out.input_path = input_path;
  out.num_cameras = p.num_cameras();
  out.num_landmarks = p.num_landmarks();
  out.num_observations = p.num_observations();
  if (compute_sparsity) {  // compute_rcs_sparsity, bal_problem.cpp:748-814: share of empty blocks of the reduced camera system
    const size_t nc = (size_t)p.num_cameras();
    std::vector<bool> mask(nc * nc, false);
    for (const auto& lm : p.landmarks())
      for (const auto& oi : lm.obs)
        for (const auto& oj : lm.obs) {
          if (oj.first < oi.first) mask[(size_t)oi.first * nc + oj.first] = true;
          else break;  // ordered map
        }
    const double nnz = (double)nc + 2.0 * (double)std::count(mask.begin(), mask.end(), true);
    out.rcs_sparsity = 1.0 - nnz / ((double)nc * (double)nc);
  }
  double sum = 0, mn = std::numeric_limits<double>::infinity(), mx = 0;
  for (const auto& lm : p.landmarks()) {
    const double k = (double)lm.obs.size();
    sum += k;
    mn = std::min(mn, k);
    mx = std::max(mx, k);
  }
  const double n = std::max(1, p.num_landmarks()), mean = sum / n;
  double var = 0;
  for (const auto& lm : p.landmarks()) var += ((double)lm.obs.size() - mean) * ((double)lm.obs.size() - mean);
  out.per_lm_obs = {mean, p.num_landmarks() ? mn : 0.0, mx, std::sqrt(var / n)};
  out.per_host_lms = DatasetSummary::Stats();
}

namespace {

std::string json_escape(const std::string& s) {
  std::string o;
  for (char ch : s) {
    if (ch == '"' || ch == '\\') { o += '\\'; o += ch; }
    else if (ch == '\n') o += "\\n";
    else o += ch;
  }
  return o;
}

// one iteration of the log after log_summary(BaLog::BaIteration&, prev, IterationSummary) (ba_log_utils.cpp:99-167):
// an unsuccessful iteration repeats the previous iteration's cost values ("for monotonic plots")
struct LoggedIteration {
  const IterationSummary* it;
  long num_obs, num_obs_valid, num_obs_valid_change;
  double cost, cost_change, cost_valid, cost_valid_change, cost_avg_valid, cost_avg_valid_change;
  double residual_block_mean, residual_block_valid_mean, grad_max_norm, grad_norm, step_norm, relative_decrease;
};

}  // namespace

void save_ba_log_json(const BalPipelineSummary& ps, const SolverOptions& o) {
  if (o.log.disable_all || o.log.log_path.empty()) return;
  const SolverSummary& s = ps.solver;
  std::vector<LoggedIteration> L;
  L.reserve(s.iterations.size());
  for (const IterationSummary& it : s.iterations) {
    LoggedIteration l{};
    l.it = &it;
    if (it.step_is_successful || L.empty()) {
      l.num_obs = it.cost.all.num_obs;
      l.num_obs_valid = it.cost.valid.num_obs;
      l.num_obs_valid_change = it.cost_change_valid.num_obs;
      l.cost = it.cost.all.error;
      l.cost_change = it.cost_change_all.error;
      l.cost_valid = it.cost.valid.error;
      l.cost_valid_change = it.cost_change_valid.error;
      l.cost_avg_valid = it.cost.valid.error_avg();
      l.cost_avg_valid_change = it.cost_change_valid.error_avg;
      l.residual_block_mean = it.cost.all.residual_mean();
      l.residual_block_valid_mean = it.cost.valid.residual_mean();
      l.grad_max_norm = it.gradient_max_norm;
      l.grad_norm = it.gradient_norm;
      l.step_norm = it.step_norm;
      l.relative_decrease = it.relative_decrease;
    } else {
      const LoggedIteration& p = L.back();
      l.num_obs = p.num_obs;
      l.num_obs_valid = p.num_obs_valid;
      l.cost = p.cost;
      l.cost_valid = p.cost_valid;
      l.cost_avg_valid = p.cost_avg_valid;
      l.residual_block_mean = p.residual_block_mean;
      l.residual_block_valid_mean = p.residual_block_valid_mean;
      l.grad_max_norm = p.grad_max_norm;
      l.grad_norm = p.grad_norm;
    }
    L.push_back(l);
  }
  FILE* f = std::fopen(o.log.log_path.c_str(), "w");
  if (!f) {
    std::fprintf(stderr, "Could not save BA log to %s.\n", o.log.log_path.c_str());
    return;
  }
  // nlohmann::json objects are ordered maps: keys come out sorted, "_static" and "_type" first
  const DatasetSummary& d = ps.dataset;
  auto stats = [&](const char* name, const DatasetSummary::Stats& st, const char* tail) {
    std::fprintf(f, "            \"%s\": {\"max\": %.17g, \"mean\": %.17g, \"min\": %.17g, \"stddev\": %.17g}%s\n", name, st.max,
                 st.mean, st.min, st.stddev, tail);
  };
  std::fprintf(f, "{\n    \"_static\": {\n        \"problem_info\": {\n");
  std::fprintf(f, "            \"input_path\": \"%s\",\n            \"num_cameras\": %d,\n            \"num_landmarks\": %d,\n"
                  "            \"num_observations\": %ld,\n", json_escape(d.input_path).c_str(), d.num_cameras, d.num_landmarks,
               d.num_observations);
  stats("per_host_lms", d.per_host_lms, ",");
  stats("per_lm_obs", d.per_lm_obs, ",");
  std::fprintf(f, "            \"rcs_sparsity\": %.17g,\n            \"type\": \"%s\"\n        },\n", d.rcs_sparsity, d.type.c_str());
  std::fprintf(f, "        \"solver\": {\n"
                  "            \"fraction_grouped\": %.17g,\n            \"grouping_time_in_seconds\": %.17g,\n"
                  "            \"jacobian_evaluation_time_in_seconds\": %.17g,\n            \"linear_solver_time_in_seconds\": %.17g,\n"
                  "            \"logging_time_in_seconds\": %.17g,\n            \"merge_factor\": true,\n"
                  "            \"message\": \"%s\",\n            \"minimizer_time_in_seconds\": %.17g,\n"
                  "            \"num_jacobian_evaluations\": %d,\n            \"num_linear_solves\": %d,\n"
                  "            \"num_residual_evaluations\": %d,\n            \"num_successful_steps\": %d,\n"
                  "            \"num_threads_available\": %d,\n            \"num_threads_given\": %d,\n"
                  "            \"num_threads_used\": %d,\n            \"num_unsuccessful_steps\": %d,\n"
                  "            \"postprocessor_time_in_seconds\": %.17g,\n            \"preprocessor_time_in_seconds\": %.17g,\n"
                  "            \"residual_evaluation_time_in_seconds\": %.17g,\n            \"resident_memory_peak\": %llu,\n"
                  "            \"solver_type\": \"%s\",\n            \"termination_type\": \"%s\",\n"
                  "            \"total_time_in_seconds\": %.17g\n        },\n",
               s.fraction_grouped, s.grouping_time_in_seconds, s.jacobian_evaluation_time_in_seconds,
               s.linear_solver_time_in_seconds, s.logging_time_in_seconds, json_escape(s.message).c_str(),
               s.minimizer_time_in_seconds, s.num_jacobian_evaluations, s.num_linear_solves, s.num_residual_evaluations,
               s.num_successful_steps, s.num_threads_available, s.num_threads_given, s.num_threads_used,
               s.num_unsuccessful_steps, s.postprocessor_time_in_seconds, s.preprocessor_time_in_seconds,
               s.residual_evaluation_time_in_seconds, s.resident_memory_peak, s.solver_type.c_str(),
               s.termination_type == CONVERGENCE ? "CONVERGENCE" : s.termination_type == NO_CONVERGENCE ? "NO_CONVERGENCE" : "FAILURE",
               s.total_time_in_seconds);
  const PipelineTimingSummary& t = ps.timing;
  std::fprintf(f, "        \"timing\": {\"load\": %.17g, \"optimize\": %.17g, \"postprocess\": %.17g, \"preprocess\": %.17g, "
                  "\"total\": %.17g}\n    },\n    \"_type\": \"rootba_povar\"",
               t.load_time, t.optimize_time, t.postprocess_time, t.preprocess_time,
               t.load_time + t.preprocess_time + t.optimize_time);  // PipelineTiming::update_total, ba_log.hpp:95-100
  auto arr = [&](const char* name, auto fmt_one) {
    std::fprintf(f, ",\n    \"%s\": [", name);
    for (size_t i = 0; i < L.size(); ++i) {
      if (i) std::fprintf(f, ", ");
      fmt_one(L[i]);
    }
    std::fprintf(f, "]");
  };
#define ARR_D(name, expr) arr(name, [&](const LoggedIteration& l) { std::fprintf(f, "%.17g", (double)(expr)); })
#define ARR_I(name, expr) arr(name, [&](const LoggedIteration& l) { std::fprintf(f, "%lld", (long long)(expr)); })
#define ARR_B(name, expr) arr(name, [&](const LoggedIteration& l) { std::fprintf(f, "%s", (expr) ? "true" : "false"); })
  // BaLog::BaIteration (ba_log.hpp:147-245), alphabetical like the reference's output
  ARR_D("back_substitution_time", l.it->back_substitution_time_in_seconds);
  ARR_D("compute_gradient_time", l.it->compute_gradient_time_in_seconds);
  ARR_D("compute_preconditioner_time", l.it->compute_preconditioner_time_in_seconds);
  ARR_D("cost", l.cost);
  ARR_D("cost_avg_valid", l.cost_avg_valid);
  ARR_D("cost_avg_valid_change", l.cost_avg_valid_change);
  ARR_D("cost_change", l.cost_change);
  ARR_D("cost_valid", l.cost_valid);
  ARR_D("cost_valid_change", l.cost_valid_change);
  ARR_D("cumulative_time", l.it->cumulative_time_in_seconds);
  ARR_D("grad_max_norm", l.grad_max_norm);
  ARR_D("grad_norm", l.grad_norm);
  ARR_D("grad_projected_max_norm", 0.0);  // never assigned by log_summary: stays at its initialiser
  ARR_D("grad_projected_norm", 0.0);
  ARR_I("iteration", l.it->iteration);
  ARR_D("iteration_time", l.it->iteration_time_in_seconds);
  ARR_D("jacobian_evaluation_time", l.it->jacobian_evaluation_time_in_seconds);
  ARR_D("landmark_damping_time", l.it->landmark_damping_time_in_seconds);
  ARR_I("linear_solver_iterations", l.it->linear_solver_iterations);
  arr("linear_solver_type", [&](const LoggedIteration& l) { std::fprintf(f, "\"%s\"", json_escape(l.it->linear_solver_type).c_str()); });
  ARR_D("logging_time", l.it->logging_time_in_seconds);
  ARR_I("num_obs", l.num_obs);
  ARR_I("num_obs_valid", l.num_obs_valid);
  ARR_I("num_obs_valid_change", l.num_obs_valid_change);
  ARR_D("perform_qr_time", l.it->perform_qr_time_in_seconds);
  ARR_D("prepare_time", l.it->prepare_time_in_seconds);
  ARR_D("relative_decrease", l.relative_decrease);
  ARR_I("resident_memory", l.it->resident_memory);
  ARR_I("resident_memory_peak", l.it->resident_memory_peak);
  ARR_D("residual_block_mean", l.residual_block_mean);
  ARR_D("residual_block_valid_mean", l.residual_block_valid_mean);
  ARR_D("residual_evaluation_time", l.it->residual_evaluation_time_in_seconds);
  ARR_D("scale_landmark_jacobian_time", l.it->scale_landmark_jacobian_time_in_seconds);
  ARR_D("scale_pose_jacobian_time", l.it->scale_pose_jacobian_time_in_seconds);
  ARR_D("solve_reduced_system_time", l.it->solve_reduced_system_time_in_seconds);
  ARR_D("stage1_time", l.it->stage1_time_in_seconds);
  ARR_D("stage2_time", l.it->stage2_time_in_seconds);
  ARR_B("step_is_nonmonotonic", l.it->step_is_nonmonotonic);
  ARR_B("step_is_successful", l.it->step_is_successful);
  ARR_B("step_is_valid", l.it->step_is_valid);
  ARR_D("step_norm", l.step_norm);
  ARR_D("step_solver_time", l.it->step_solver_time_in_seconds);
  ARR_D("trust_region_radius", l.it->trust_region_radius);
  ARR_D("update_cameras_time", l.it->update_cameras_time_in_seconds);
#undef ARR_D
#undef ARR_I
#undef ARR_B
  std::fprintf(f, "\n}\n");
  std::fclose(f);
  if (!o.log.disable_all) std::fprintf(stderr, "Saved log for %zu iterations to %s.\n", L.size(), o.log.log_path.c_str());
}

}  // namespace povar_host
