// The `bal` executable of the drop-in surface (src/app/bal.cpp:44-103): parse -> load ->
// bundle_adjust_manual -> log.
#include <cstdio>

#include "linearizor.hpp"

int main(int argc, char** argv) {
  using namespace povar_host;
  BalAppOptions options;
  if (!parse_bal_app_arguments(argc, argv, options)) return 1;
  // src/app/bal.cpp:83-99: load (+ dataset summary, timing) -> solve -> postprocess -> log
  BalPipelineSummary summary;
  BalProblem bal_problem = load_normalized_bal_problem(options.dataset, &summary.dataset, &summary.timing);
  bundle_adjust_manual(bal_problem, options.solver, &summary.solver, &summary.timing);
  save_ba_log_json(summary, options.solver);
  return 0;
}
