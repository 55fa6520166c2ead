// The reference's operator interface for this path, restated (solver/linearizor.hpp:48-82,
// solver/solver_summary.hpp, bal/residual_info.hpp:59-102).  VecX is a std::vector<double>
// (the reference's Eigen::VectorXd).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "bal_problem.hpp"
#include "solver_options.hpp"

namespace povar_host {

struct ResidualItem {  // residual_info.hpp:59-75
  long num_obs = 0;
  double error = 0;
  double residual_sum = 0;
  double error_avg() const { return num_obs > 0 ? error / num_obs : 0.0; }
  double residual_mean() const { return num_obs > 0 ? residual_sum / num_obs : 0.0; }
};
struct ResidualInfo {  // residual_info.hpp:78-92
  ResidualItem all, valid;
  bool is_numerically_valid = true;
};
std::string error_summary_oneline(const ResidualInfo& info, bool valid_first);

struct ResidualChangeItem {  // residual_info.hpp: "previous - this" (residual_info.cpp:43-53)
  long num_obs = 0;
  double error = 0, error_avg = 0;
};
struct IterationSummary {  // solver_summary.hpp:120-222
  int iteration = 0;
  bool step_is_valid = false;
  bool step_is_nonmonotonic = false;
  bool step_is_successful = false;
  ResidualInfo cost;
  ResidualChangeItem cost_change_all, cost_change_valid;
  double cost_change_all_error = 0;
  double gradient_max_norm = 0, gradient_norm = 0, step_norm = 0;  // never set on this path (logged as 0)
  double logging_time_in_seconds = 0, perform_qr_time_in_seconds = 0, compute_preconditioner_time_in_seconds = 0,
         compute_gradient_time_in_seconds = 0;
  unsigned long long resident_memory = 0, resident_memory_peak = 0;
  double relative_decrease = 0;
  double trust_region_radius = 0;
  int linear_solver_iterations = 0;
  std::string linear_solver_message;
  std::string linear_solver_type;
  double iteration_time_in_seconds = 0, cumulative_time_in_seconds = 0, step_solver_time_in_seconds = 0;
  double residual_evaluation_time_in_seconds = 0, jacobian_evaluation_time_in_seconds = 0;
  double scale_landmark_jacobian_time_in_seconds = 0, stage1_time_in_seconds = 0;
  double scale_pose_jacobian_time_in_seconds = 0, landmark_damping_time_in_seconds = 0;
  double stage2_time_in_seconds = 0, prepare_time_in_seconds = 0;
  double solve_reduced_system_time_in_seconds = 0, back_substitution_time_in_seconds = 0;
  double update_cameras_time_in_seconds = 0;
};

enum TerminationType { CONVERGENCE, NO_CONVERGENCE, FAILURE };

struct SolverSummary {
  std::string solver_type;
  std::vector<IterationSummary> iterations;
  ResidualInfo initial_cost, final_cost;
  TerminationType termination_type = NO_CONVERGENCE;
  std::string message;
  int num_successful_steps = 0, num_unsuccessful_steps = 0;
  int num_linear_solves = 0, num_residual_evaluations = 0, num_jacobian_evaluations = 0;
  double preprocessor_time_in_seconds = 0, minimizer_time_in_seconds = 0, total_time_in_seconds = 0;
  double linear_solver_time_in_seconds = 0, residual_evaluation_time_in_seconds = 0,
         jacobian_evaluation_time_in_seconds = 0;
  double logging_time_in_seconds = 0, grouping_time_in_seconds = 0, postprocessor_time_in_seconds = 0;
  int num_threads_given = 0, num_threads_used = 0, num_threads_available = 0;
  unsigned long long resident_memory_peak = 0;
  double fraction_grouped = 0;
};

// bal/bal_pipeline_summary.hpp:42-79
struct DatasetSummary {
  struct Stats { double mean = 0, min = 0, max = 0, stddev = 0; };
  std::string type, input_path;
  int num_cameras = 0, num_landmarks = 0;
  long num_observations = 0;
  double rcs_sparsity = 0;
  Stats per_lm_obs, per_host_lms;
};
struct PipelineTimingSummary { double load_time = 0, preprocess_time = 0, optimize_time = 0, postprocess_time = 0; };
struct BalPipelineSummary {
  DatasetSummary dataset;
  PipelineTimingSummary timing;
  SolverSummary solver;
};

class Linearizor {
 public:
  using VecX = std::vector<double>;
  // factories, solver/linearizor.cpp:47-80: POWER_VARPROJ / POWER_SCHUR_COMPLEMENT / RIPOBA select the
  // power-series linearizor (LinearizorPowerVarproj), PCG / CHOLESKY / RIPCG the explicit-Schur-complement
  // one (LinearizorSC); both are served by the MI355X library behind include/povar_hip.h.
  static std::unique_ptr<Linearizor> create(BalProblem& bal_problem, const SolverOptions& options,
                                            SolverSummary* summary = nullptr);
  static std::unique_ptr<Linearizor> create_homogeneous(BalProblem& bal_problem, const SolverOptions& options,
                                                        SolverSummary* summary = nullptr);
  virtual ~Linearizor() = default;

  virtual void start_iteration(IterationSummary* it_summary = nullptr) = 0;
  virtual void compute_error_pOSE(ResidualInfo& ri, bool initialization_varproj) = 0;
  virtual void compute_error_homogeneous(ResidualInfo& ri, bool initialization_varproj) = 0;
  virtual void initialize_varproj_lm_pOSE(double alpha, bool initialization_varproj) = 0;
  virtual void linearize_pOSE(double alpha) = 0;
  virtual void linearize_projective_space_homogeneous() = 0;
  virtual VecX solve_joint(double lambda, double relative_error_change) = 0;
  virtual VecX solve(const SolverOptions& solver_options, double lambda, double relative_error_change) = 0;
  virtual double apply_joint(VecX&& inc) = 0;
  virtual double apply(const SolverOptions& solver_options, double alpha, VecX&& inc) = 0;
  virtual void finish_iteration() = 0;
};

// hook for alternative implementations (the tests register an oracle-backed one)
using LinearizorFactory = std::unique_ptr<Linearizor> (*)(BalProblem&, const SolverOptions&, SolverSummary*, bool homogeneous);
void set_linearizor_factory(LinearizorFactory f);

// solver/bal_bundle_adjustment.cpp:848-876
void bundle_adjust_manual(BalProblem& bal_problem, const SolverOptions& solver_options,
                          SolverSummary* output_solver_summary = nullptr, PipelineTimingSummary* output_timing = nullptr);
// BaLog::save_json (bal/ba_log.cpp:63-150) after log_summary (bal/ba_log_utils.cpp:43-186): every per-iteration
// field of BaLog::BaIteration as a flat array, _static {problem_info, timing, solver}, _type
void save_ba_log_json(const BalPipelineSummary& summary, const SolverOptions& options);
// BalProblem::summarize_problem (bal/bal_problem.cpp:817-859)
void summarize_problem(const BalProblem& problem, const std::string& input_path, bool compute_sparsity, DatasetSummary& out);
// util/system_utils.cpp:52-95 (Linux branch)
bool get_memory_info(unsigned long long& resident, unsigned long long& resident_peak);

}  // namespace povar_host
