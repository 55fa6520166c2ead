#include <chrono>
#include "bal_problem.hpp"

#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>

#include "solver_options.hpp"
#include "linearizor.hpp"

namespace povar_host {

namespace {
[[noreturn]] void fatal(const std::string& m) {
  std::fprintf(stderr, "FATAL: %s\n", m.c_str());
  std::exit(1);
}
void scan_int(FILE* f, int* v) {
  if (std::fscanf(f, "%d", v) != 1) throw std::runtime_error("parse");
}
void scan_dbl(FILE* f, double* v) {
  if (std::fscanf(f, "%lf", v) != 1) throw std::runtime_error("parse");
}
}  // namespace

long BalProblem::num_observations() const {
  long n = 0;
  for (const auto& l : landmarks_) n += (long)l.obs.size();
  return n;
}

// load_bal_eccv, bal_problem.cpp:183-303 (the PoVar data_custom format)
void BalProblem::load_bal_eccv(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) fatal("Could not open '" + path + "'");
  try {
    int nc, nl, no;
    scan_int(f, &nc); scan_int(f, &nl); scan_int(f, &no);
    if (nc <= 0 || nl <= 0 || no <= 0) throw std::runtime_error("header");
    cameras_.assign(nc, Camera());
    landmarks_.assign(nl, Landmark());
    for (int i = 0; i < no; ++i) {
      int c, l;
      scan_int(f, &c); scan_int(f, &l);
      if (c < 0 || c >= nc || l < 0 || l >= nl) throw std::runtime_error("index");
      auto ins = landmarks_[l].obs.emplace(c, std::array<double, 2>{});
      if (!ins.second) fatal("Invalid file '" + path + "'");  // duplicate pair, bal_problem.cpp:227
      double u, v;
      scan_dbl(f, &u); scan_dbl(f, &v);
      ins.first->second = {u, -v};  // invert y axis, bal_problem.cpp:240
    }
    for (int i = 0; i < nc; ++i) {
      double p[15];
      for (double& x : p) scan_dbl(f, &x);
      for (int k = 0; k < 12; ++k) cameras_[i].space_matrix[k] = p[k];  // bal_problem.cpp:251-253
      cameras_[i].intrinsics = {p[12], p[13], p[14]};
    }
    // file landmarks are replaced by N(0,1) draws in the reference (bal_problem.cpp:261-267) and
    // then overwritten by the VarPro initialisation (linearizor_base.cpp:64-65): keep the file values.
    for (int i = 0; i < nl; ++i)
      for (int k = 0; k < 3; ++k) scan_dbl(f, &landmarks_[i].p_w[k]);
  } catch (const std::exception&) {
    fatal("Failed to parse '" + path + "'");
  }
  std::fclose(f);
  if (!quiet)
    std::fprintf(stderr, "Loaded BAL problem (%d cams, %d lms, %ld obs) from '%s'\n", num_cameras(),
                 num_landmarks(), num_observations(), path.c_str());
}

// --create-dataset: original BAL file -> data_custom/<name> with random initial cameras
// (bal_problem.cpp:307-471).  The reference seeds std::mt19937 from random_device; here the seed
// is --random-seed so the written file is reproducible.
void BalProblem::load_bal_varproj_space_matrix_write(const std::string& path, int seed) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) fatal("Could not open '" + path + "'");
  ::mkdir("data_custom", 0755);
  const size_t found = path.find_last_of('/');
  const std::string out_name = "data_custom/" + path.substr(found == std::string::npos ? 0 : found + 1);
  FILE* o = std::fopen(out_name.c_str(), "w");
  if (!o) fatal("Could not open '" + out_name + "'");
  std::mt19937 gen((unsigned)seed);
  try {
    int nc, nl, no;
    scan_int(f, &nc); scan_int(f, &nl); scan_int(f, &no);
    std::fprintf(o, "%d %d %d", nc, nl, no);
    cameras_.assign(nc, Camera());
    landmarks_.assign(nl, Landmark());
    for (int i = 0; i < no; ++i) {
      int c, l;
      double u, v;
      scan_int(f, &c); scan_int(f, &l); scan_dbl(f, &u); scan_dbl(f, &v);
      if (c < 0 || c >= nc || l < 0 || l >= nl) throw std::runtime_error("index");
      std::fprintf(o, "\n%d %d %lf %lf", c, l, u, v);
      auto ins = landmarks_[l].obs.emplace(c, std::array<double, 2>{u, -v});
      if (!ins.second) fatal("Invalid file '" + path + "'");
    }
    for (int i = 0; i < nc; ++i) {
      std::normal_distribution<double> d(0, 1);
      double in9[9], p[15];
      for (double& x : in9) scan_dbl(f, &x);
      for (double& x : p) x = d(gen);  // bal_problem.cpp:398-400
      auto& P = cameras_[i].space_matrix;
      for (int k = 0; k < 8; ++k) P[k] = p[k];
      P[8] = 0; P[9] = 0; P[10] = 0; P[11] = 1;  // bal_problem.cpp:404-407
      for (int k = 0; k < 12; ++k) std::fprintf(o, "\n%lf", P[k]);
      cameras_[i].intrinsics = {in9[6], in9[7], in9[8]};
      for (int k = 6; k < 9; ++k) std::fprintf(o, "\n%lf", in9[k]);
    }
    for (int i = 0; i < nl; ++i)
      for (int k = 0; k < 3; ++k) {
        scan_dbl(f, &landmarks_[i].p_w[k]);
        std::fprintf(o, "\n%lf", landmarks_[i].p_w[k]);
      }
  } catch (const std::exception&) {
    fatal("Failed to parse '" + path + "'");
  }
  std::fclose(o);
  std::fclose(f);
  if (!quiet) std::fprintf(stderr, "Wrote '%s'\n", out_name.c_str());
}

void BalProblem::backup_pOSE() {
  if (mirror) return mirror->backup_pOSE();
  for (auto& c : cameras_) c.space_matrix_backup = c.space_matrix;
  for (auto& l : landmarks_) l.p_w_backup = l.p_w;
}
void BalProblem::restore_pOSE() {
  if (mirror) return mirror->restore_pOSE();
  for (auto& c : cameras_) c.space_matrix = c.space_matrix_backup;
  for (auto& l : landmarks_) l.p_w = l.p_w_backup;
}
void BalProblem::backup_joint() {
  if (mirror) return mirror->backup_joint();
  for (auto& c : cameras_) c.space_matrix_backup = c.space_matrix;
  for (auto& l : landmarks_) l.p_w_homogeneous_backup = l.p_w_homogeneous;
}
void BalProblem::restore_joint() {
  if (mirror) return mirror->restore_joint();
  for (auto& c : cameras_) c.space_matrix = c.space_matrix_backup;
  for (auto& l : landmarks_) l.p_w_homogeneous = l.p_w_homogeneous_backup;
}

void BalProblem::flatten(std::vector<int>& lm_off, std::vector<int>& cam_idx, std::vector<double>& obs) const {
  lm_off.assign(1, 0);
  cam_idx.clear();
  obs.clear();
  for (const auto& l : landmarks_) {
    for (const auto& kv : l.obs) {  // std::map: ascending camera index (landmark_block.hpp:104-108)
      cam_idx.push_back(kv.first);
      obs.push_back(kv.second[0]);
      obs.push_back(kv.second[1]);
    }
    lm_off.push_back((int)cam_idx.size());
  }
}

// load_normalized_bal_problem, bal_problem.cpp:874-955.  normalize/perturb/filter_obs have no
// effect on the PoVar state and are not restated.
BalProblem load_normalized_bal_problem(const BalDatasetOptions& options, DatasetSummary* dataset_summary,
                                       PipelineTimingSummary* timing_summary) {
  const auto t0 = std::chrono::steady_clock::now();
  BalProblem p;
  p.quiet = options.quiet;
  if (options.create_dataset) {
    p.load_bal_varproj_space_matrix_write(options.input, options.random_seed);
    std::exit(0);  // bal_problem.cpp:899-903
  }
  p.load_bal_eccv(options.input);
  const double time_load = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (timing_summary) {  // bal_problem.cpp:939-942 (the preprocessing steps are no-ops on this path)
    timing_summary->load_time = time_load;
    timing_summary->preprocess_time = 0;
  }
  if (dataset_summary) summarize_problem(p, options.input, true, *dataset_summary);  // :944-947
  return p;
}

}  // namespace povar_host
