// Option structs of the drop-in surface: same names and defaults as the reference
// (bal/solver_options.hpp:95-305, bal/bal_residual_options.hpp:53-62, bal/bal_dataset_options.hpp:49-94,
// bal/ba_log_options.hpp:46-56).  The reference's reflection/clipp/TOML option framework is out of
// scope; a flat parser reproduces the flag names (cli.cpp).
#pragma once
#include <string>

namespace povar_host {

struct BalResidualOptions {
  enum class RobustNorm { NONE, HUBER, CAUCHY };
  RobustNorm robust_norm = RobustNorm::NONE;
  double huber_parameter = 1.0;
};

struct BaLogOptions {
  std::string log_path = "ba_log.json";
  bool disable_all = false;
};

struct SolverOptions {
  enum class SolverType { PCG, POWER_SCHUR_COMPLEMENT, POWER_VARPROJ, CHOLESKY };
  enum class SolverTypeRiemannian { RIPOBA, RIPCG };
  enum class OptimizedCost { ERROR, ERROR_VALID, ERROR_VALID_AVG };
  enum class PreconditionerType { JACOBI, SCHUR_JACOBI };

  SolverType solver_type_step_1 = SolverType::POWER_VARPROJ;
  SolverTypeRiemannian solver_type_step_2 = SolverTypeRiemannian::RIPOBA;
  int verbosity_level = 2;
  bool debug = false;
  int num_threads = 0;
  BalResidualOptions residual;
  double alpha = 0.01;
  BaLogOptions log;
  OptimizedCost optimized_cost = OptimizedCost::ERROR;
  int max_num_iterations_step_1 = 50;
  int max_num_iterations_step_2 = 50;
  double min_relative_decrease = 0.0;
  double initial_trust_region_radius = 1e4;
  double min_trust_region_radius = 1e-32;
  double max_trust_region_radius = 1e16;
  int min_linear_solver_iterations = 0;    // bal/solver_options.hpp:196-200
  int max_linear_solver_iterations = 500;  // bal/solver_options.hpp:201-203
  double eta = 1e-2;
  double r_tolerance = -1.0;
  bool jacobi_scaling = true;
  double jacobi_scaling_epsilon = 0.0;
  PreconditionerType preconditioner_type = PreconditionerType::SCHUR_JACOBI;
  double function_tolerance = 1e-6;
  int power_sc_iterations = 10;
  double initial_vee = 2.0;
  double vee_factor = 2.0;
  // MI355X build: which E0 operator form the device uses (not a reference option)
  std::string e0_mode = "ldsacc";  // "ldsacc" (fastest), "implicit" (bit-reproducible), "tiles" (stored tiles)
  // bit-reproducible results run to run (povar_options.flags: POVAR_FLAG_DETERMINISTIC; 1.26 x the default term on venice-1778)
  bool deterministic = false;
  int device = 0;
  // landmark shards = device contexts of ONE process (BASELINE configs 4 / 5: "landmarks sharded across 8 x MI355X"):
  // shard r runs on device (device + r) mod device count; one exchange step per power-series term (RCCL, or an in-process
  // all-reduce when shards share a device)
  int gpus = 1;

  // solver_options.cpp:41-51
  bool use_projection_validity_check() const { return optimized_cost != OptimizedCost::ERROR; }
};

struct BalDatasetOptions {
  std::string input;
  bool create_dataset = false;
  int random_seed = 38401;
  bool quiet = false;
};

struct BalAppOptions {
  BalDatasetOptions dataset;
  SolverOptions solver;
};

// returns false (after printing a message) on a parse error or --help
bool parse_bal_app_arguments(int argc, char** argv, BalAppOptions& options);
const char* to_string(SolverOptions::SolverType t);
const char* to_string(SolverOptions::SolverTypeRiemannian t);

}  // namespace povar_host
