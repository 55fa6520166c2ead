// Flat CLI parser reproducing the reference's flag names: "--" + nested prefix + member name with
// '_' -> '-' (cli/cli_options.cpp:61, 134), booleans as --x / --no-x (cli_options.cpp:85-91), enums
// from strings (cli_options.cpp:102-111).  Prefixes: dataset and solver members have none, the
// nested solver.residual and solver.log structs use "residual-" and "log-".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>

#include "solver_options.hpp"

namespace povar_host {

const char* to_string(SolverOptions::SolverType t) {
  switch (t) {
    case SolverOptions::SolverType::PCG: return "PCG";
    case SolverOptions::SolverType::POWER_SCHUR_COMPLEMENT: return "POWER_SCHUR_COMPLEMENT";
    case SolverOptions::SolverType::POWER_VARPROJ: return "POWER_VARPROJ";
    default: return "CHOLESKY";
  }
}
const char* to_string(SolverOptions::SolverTypeRiemannian t) {
  return t == SolverOptions::SolverTypeRiemannian::RIPOBA ? "RIPOBA" : "RIPCG";
}

namespace {
template <class E>
bool parse_enum(const std::string& v, const std::map<std::string, E>& m, E& out) {
  auto it = m.find(v);
  if (it == m.end()) return false;
  out = it->second;
  return true;
}
}  // namespace

bool parse_bal_app_arguments(int argc, char** argv, BalAppOptions& o) {
  using S = SolverOptions;
  std::map<std::string, std::function<bool(const std::string&)>> val;
  std::map<std::string, bool*> flag;
  auto dbl = [&](const char* n, double* p) { val[n] = [p](const std::string& v) { char* e; *p = std::strtod(v.c_str(), &e); return *e == 0; }; };
  auto integer = [&](const char* n, int* p) { val[n] = [p](const std::string& v) { char* e; *p = (int)std::strtol(v.c_str(), &e, 10); return *e == 0; }; };
  auto str = [&](const char* n, std::string* p) { val[n] = [p](const std::string& v) { *p = v; return true; }; };
  str("input", &o.dataset.input);
  integer("random-seed", &o.dataset.random_seed);
  flag["create-dataset"] = &o.dataset.create_dataset;
  flag["quiet"] = &o.dataset.quiet;
  val["solver-type-step-1"] = [&](const std::string& v) {
    return parse_enum<S::SolverType>(v, {{"PCG", S::SolverType::PCG}, {"POWER_SCHUR_COMPLEMENT", S::SolverType::POWER_SCHUR_COMPLEMENT},
                                         {"POWER_VARPROJ", S::SolverType::POWER_VARPROJ}, {"CHOLESKY", S::SolverType::CHOLESKY}}, o.solver.solver_type_step_1); };
  val["solver-type-step-2"] = [&](const std::string& v) {
    return parse_enum<S::SolverTypeRiemannian>(v, {{"RIPOBA", S::SolverTypeRiemannian::RIPOBA}, {"RIPCG", S::SolverTypeRiemannian::RIPCG}}, o.solver.solver_type_step_2); };
  val["optimized-cost"] = [&](const std::string& v) {
    return parse_enum<S::OptimizedCost>(v, {{"ERROR", S::OptimizedCost::ERROR}, {"ERROR_VALID", S::OptimizedCost::ERROR_VALID},
                                            {"ERROR_VALID_AVG", S::OptimizedCost::ERROR_VALID_AVG}}, o.solver.optimized_cost); };
  val["preconditioner-type"] = [&](const std::string& v) {
    return parse_enum<S::PreconditionerType>(v, {{"JACOBI", S::PreconditionerType::JACOBI}, {"SCHUR_JACOBI", S::PreconditionerType::SCHUR_JACOBI}}, o.solver.preconditioner_type); };
  val["residual-robust-norm"] = [&](const std::string& v) {
    using R = BalResidualOptions::RobustNorm;
    return parse_enum<R>(v, {{"NONE", R::NONE}, {"HUBER", R::HUBER}, {"CAUCHY", R::CAUCHY}}, o.solver.residual.robust_norm); };
  dbl("residual-huber-parameter", &o.solver.residual.huber_parameter);
  str("log-log-path", &o.solver.log.log_path);
  flag["log-disable-all"] = &o.solver.log.disable_all;
  integer("verbosity-level", &o.solver.verbosity_level);
  flag["debug"] = &o.solver.debug;
  integer("num-threads", &o.solver.num_threads);
  dbl("alpha", &o.solver.alpha);
  integer("max-num-iterations-step-1", &o.solver.max_num_iterations_step_1);
  integer("max-num-iterations-step-2", &o.solver.max_num_iterations_step_2);
  dbl("min-relative-decrease", &o.solver.min_relative_decrease);
  dbl("initial-trust-region-radius", &o.solver.initial_trust_region_radius);
  dbl("min-trust-region-radius", &o.solver.min_trust_region_radius);
  dbl("max-trust-region-radius", &o.solver.max_trust_region_radius);
  integer("min-linear-solver-iterations", &o.solver.min_linear_solver_iterations);
  integer("max-linear-solver-iterations", &o.solver.max_linear_solver_iterations);
  dbl("eta", &o.solver.eta);
  dbl("r-tolerance", &o.solver.r_tolerance);
  flag["jacobi-scaling"] = &o.solver.jacobi_scaling;
  dbl("jacobi-scaling-epsilon", &o.solver.jacobi_scaling_epsilon);
  dbl("function-tolerance", &o.solver.function_tolerance);
  integer("power-sc-iterations", &o.solver.power_sc_iterations);
  dbl("initial-vee", &o.solver.initial_vee);
  dbl("vee-factor", &o.solver.vee_factor);
  str("e0-mode", &o.solver.e0_mode);
  flag["deterministic"] = &o.solver.deterministic;
  integer("device", &o.solver.device);
  integer("gpus", &o.solver.gpus);

  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "-h" || a == "--help") {
      std::printf("usage: bal --input FILE [--create-dataset] [options]\n  value options:");
      for (auto& kv : val) std::printf(" --%s", kv.first.c_str());
      std::printf("\n  boolean options (--x / --no-x):");
      for (auto& kv : flag) std::printf(" --%s", kv.first.c_str());
      std::printf("\n");
      return false;
    }
    if (a.rfind("--", 0) != 0) {
      std::fprintf(stderr, "unexpected argument '%s'\n", a.c_str());
      return false;
    }
    a = a.substr(2);
    std::string v;
    bool has_v = false;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); has_v = true; }
    if (flag.count(a)) { *flag[a] = true; continue; }
    if (a.rfind("no-", 0) == 0 && flag.count(a.substr(3))) { *flag[a.substr(3)] = false; continue; }
    auto it = val.find(a);
    if (it == val.end()) { std::fprintf(stderr, "unknown option '--%s'\n", a.c_str()); return false; }
    if (!has_v) {
      if (i + 1 >= argc) { std::fprintf(stderr, "option '--%s' needs a value\n", a.c_str()); return false; }
      v = argv[++i];
    }
    if (!it->second(v)) { std::fprintf(stderr, "invalid value '%s' for '--%s'\n", v.c_str(), a.c_str()); return false; }
  }
  if (o.dataset.input.empty()) { std::fprintf(stderr, "missing --input\n"); return false; }
  return true;
}

}  // namespace povar_host
