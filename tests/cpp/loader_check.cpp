// test-only: loads a data_custom file with the host loader (host/bal_problem.cpp) and prints what it parsed, one
// value per token in file order semantics: "obs <lm> <cam> <u> <v>" (ascending camera per landmark), "cam <15 values>",
// "lm <3 values>", all with %.17g -- tests/test_abi_and_host.py compares with Python's float().
#include <cstdio>

#include "../../povar_amd/csrc/host/bal_problem.hpp"

// bal_problem.cpp's load_normalized_bal_problem reports through summarize_problem (bal_bundle_adjustment.cpp, which drags
// the device library in): this check only loads -- a proper stub instead of letting the linker ignore what is missing
namespace povar_host {
void summarize_problem(const BalProblem&, const std::string&, bool, DatasetSummary&) {}
}  // namespace povar_host

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  povar_host::BalProblem p;
  p.quiet = true;
  p.load_bal_eccv(argv[1]);
  int l = 0;
  for (const auto& lm : p.landmarks()) {
    for (const auto& kv : lm.obs) std::printf("obs %d %d %.17g %.17g\n", l, kv.first, kv.second[0], kv.second[1]);
    ++l;
  }
  for (const auto& c : p.cameras()) {
    std::printf("cam");
    for (double v : c.space_matrix) std::printf(" %.17g", v);
    for (double v : c.intrinsics) std::printf(" %.17g", v);
    std::printf("\n");
  }
  for (const auto& lm : p.landmarks()) std::printf("lm %.17g %.17g %.17g\n", lm.p_w[0], lm.p_w[1], lm.p_w[2]);
  return 0;
}
