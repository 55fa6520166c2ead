#include <chrono>
#include "bal_problem.hpp"

#include <sys/stat.h>

#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <thread>

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

// The whole file in memory, whitespace-separated tokens parsed in place: what operator>> / fscanf accept for these files
// (venice-1778, 188 MB of text: 2.3 s of a 3.3 s `bal` run went into fscanf and the std::map nodes).
class Tokens {
 public:
  explicit Tokens(FILE* f) {
    struct stat st;
    const size_t hint = ::fstat(::fileno(f), &st) == 0 && st.st_size > 0 ? (size_t)st.st_size : (size_t)1 << 20;
    buf_.resize(hint + 1);
    size_t n = 0;
    for (;;) {
      if (n == buf_.size()) buf_.resize(buf_.size() * 2);
      const size_t got = std::fread(&buf_[n], 1, buf_.size() - n, f);
      if (got == 0) break;
      n += got;
    }
    buf_.resize(n + 1);
    buf_[n] = 0;  // strtod stops here
    p_ = buf_.data();
    end_ = p_ + n;
  }
  int next_int() {
    skip();
    const char* q = p_ < end_ && *p_ == '+' ? p_ + 1 : p_;
    int v = 0;
    const auto r = std::from_chars(q, end_, v);
    // the number must be the whole token ("12.5" where an index is expected is an error, as it is for fscanf("%d") followed
    // by the next field: consuming a part of it would shift every later field of the piece)
    if (r.ec != std::errc() || r.ptr == q || (r.ptr < end_ && (unsigned char)*r.ptr > ' ')) throw std::runtime_error("parse");
    p_ = r.ptr;
    return v;
  }
  double next_double() {
    skip();
    // Decimal literals with at most 15 significant digits and a power of ten within 10^+-22 convert exactly with one
    // multiplication or division (both operands exact in fp64: the result is the correctly rounded one strtod gives);
    // the "%lf" files of this format are all of that kind.  Everything else goes to strtod.  (libstdc++ 11's
    // from_chars<double> wraps strtod in a per-call locale switch: slower than fscanf's path and serialised across threads.)
    const char* q = p_;
    bool neg = false;
    if (q < end_ && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    unsigned long long mant = 0;
    int digits = 0, exp10 = 0;
    bool any = false;
    while (q < end_ && *q >= '0' && *q <= '9') {
      if (mant || *q != '0') { mant = mant * 10 + (unsigned)(*q - '0'); ++digits; }
      any = true;
      ++q;
    }
    if (q < end_ && *q == '.') {
      ++q;
      while (q < end_ && *q >= '0' && *q <= '9') {
        if (mant || *q != '0') { mant = mant * 10 + (unsigned)(*q - '0'); ++digits; }
        --exp10;
        any = true;
        ++q;
      }
    }
    bool simple = any && digits <= 15;
    if (simple && q < end_ && (*q == 'e' || *q == 'E')) {
      const char* r = q + 1;
      bool eneg = false;
      if (r < end_ && (*r == '-' || *r == '+')) { eneg = *r == '-'; ++r; }
      int e = 0, nd = 0;
      while (r < end_ && *r >= '0' && *r <= '9' && nd < 4) { e = e * 10 + (*r - '0'); ++r; ++nd; }
      if (nd == 0 || (r < end_ && *r >= '0' && *r <= '9')) simple = false;
      else { exp10 += eneg ? -e : e; q = r; }
    }
    if (simple && (q == end_ || (unsigned char)*q <= ' ') && exp10 >= -22 && exp10 <= 22) {
      static const double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                     1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
      const double m = (double)mant;
      const double v = exp10 < 0 ? m / p10[-exp10] : m * p10[exp10];
      p_ = q;
      return neg ? -v : v;
    }
    // everything else: strtod on the buffer itself (NUL-terminated behind end_: the constructor keeps one spare byte),
    // and the number must be the whole token
    char* e = nullptr;
    const double v = std::strtod(p_, &e);
    if (e == p_ || e > end_ || (e < end_ && (unsigned char)*e > ' ')) throw std::runtime_error("parse");
    p_ = e;
    return v;
  }

  // The next n_obs * 4 + n_tail tokens at once, on all CPUs the process may use: `cam lm u v` per observation into
  // cl[2 n_obs] / uv[2 n_obs], then n_tail doubles into tail.  Format-agnostic like the scalar path (any whitespace,
  // any line structure): every piece of the buffer counts its tokens first, a prefix sum gives each piece the index
  // of its first token.
  void bulk(size_t n_obs, size_t n_tail, std::vector<int>& cl, std::vector<double>& uv, std::vector<double>& tail) {
    cl.resize(2 * n_obs);
    uv.resize(2 * n_obs);
    tail.resize(n_tail);
    const size_t want = 4 * n_obs + n_tail;
    const int n_thr = (int)std::max<size_t>(1, std::min<size_t>(usable_cpus(), (size_t)(end_ - p_) / (1 << 20) + 1));
    auto is_ws = [](char ch) { return (unsigned char)ch <= ' '; };
    std::vector<const char*> cut(n_thr + 1);
    for (int i = 0; i <= n_thr; ++i) {
      const char* q = p_ + (size_t)(end_ - p_) * i / n_thr;
      while (q > p_ && q < end_ && !is_ws(q[-1])) ++q;  // not inside a token
      cut[i] = q;
    }
    std::vector<size_t> first(n_thr + 1, 0);
    std::vector<int> bad(n_thr, 0);
    auto run = [&](auto&& fn) {
      std::vector<std::thread> pool;
      for (int i = 1; i < n_thr; ++i) pool.emplace_back(fn, i);
      fn(0);
      for (auto& th : pool) th.join();
    };
    run([&](int i) {
      size_t n = 0;
      bool in = false;
      for (const char* q = cut[i]; q < cut[i + 1]; ++q) {
        const bool w = is_ws(*q);
        n += !w && !in;
        in = !w;
      }
      first[i + 1] = n;
    });
    for (int i = 0; i < n_thr; ++i) first[i + 1] += first[i];
    if (first[n_thr] < want) throw std::runtime_error("eof");
    run([&](int i) {
      Tokens t(cut[i], cut[i + 1]);
      try {
        for (size_t k = first[i]; k < first[i + 1] && k < want; ++k) {
          if (k < 4 * n_obs) {
            if ((k & 3) < 2) cl[2 * (k >> 2) + (k & 3)] = t.next_int();
            else uv[2 * (k >> 2) + (k & 3) - 2] = t.next_double();
          } else {
            tail[k - 4 * n_obs] = t.next_double();
          }
        }
      } catch (const std::exception&) {
        bad[i] = 1;
      }
    });
    for (int b : bad)
      if (b) throw std::runtime_error("parse");
    p_ = end_;  // (nothing this loader reads follows the landmark block)
  }

 private:
  Tokens(const char* b, const char* e) : p_(b), end_(e) {}
  static size_t usable_cpus() {  // hardware threads, cut by the cgroup CPU quota, at most 16
    size_t n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      long period = 0;
      if (std::fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0)
        n = std::min<size_t>(n, (size_t)std::max<long>(1, (std::atol(q) + period - 1) / period));
      std::fclose(f);
    }
    return std::min<size_t>(n, 16);
  }
  void skip() {
    while (p_ < end_ && (unsigned char)*p_ <= ' ') ++p_;
    if (p_ == end_) throw std::runtime_error("eof");
  }
  std::string buf_;
  const char* p_ = nullptr;
  const char* end_ = nullptr;
};
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
    Tokens t(f);
    const int nc = t.next_int(), nl = t.next_int(), no = t.next_int();
    if (nc <= 0 || nl <= 0 || no <= 0) throw std::runtime_error("header");
    cameras_.assign(nc, Camera());
    landmarks_.assign(nl, Landmark());
    std::vector<int> cl;
    std::vector<double> uv, tail;
    t.bulk((size_t)no, 15 * (size_t)nc + 3 * (size_t)nl, cl, uv, tail);
    for (int i = 0; i < no; ++i) {
      const int c = cl[2 * (size_t)i], l = cl[2 * (size_t)i + 1];
      if (c < 0 || c >= nc || l < 0 || l >= nl) throw std::runtime_error("index");
      const auto ins = landmarks_[l].obs.emplace(c, std::array<double, 2>{uv[2 * (size_t)i], -uv[2 * (size_t)i + 1]});  // invert y axis, bal_problem.cpp:240
      if (!ins.second) fatal("Invalid file '" + path + "'");  // duplicate pair, bal_problem.cpp:227
    }
    for (int i = 0; i < nc; ++i) {
      const double* p = tail.data() + 15 * (size_t)i;
      for (int k = 0; k < 12; ++k) cameras_[i].space_matrix[k] = p[k];  // bal_problem.cpp:251-253
      cameras_[i].intrinsics = {p[12], p[13], p[14]};
    }
    // file landmarks are replaced by N(0,1) draws in the reference (bal_problem.cpp:261-267) and
    // then overwritten by the VarPro initialisation (linearizor_base.cpp:64-65): keep the file values.
    for (int i = 0; i < nl; ++i)
      for (int k = 0; k < 3; ++k) landmarks_[i].p_w[k] = tail[15 * (size_t)nc + 3 * (size_t)i + k];
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
