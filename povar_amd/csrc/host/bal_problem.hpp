// BalProblem: the data model and loaders of the drop-in surface (bal/bal_problem.hpp:66-339,
// bal/bal_problem.cpp:183-471, 658-708), restated without Eigen/Sophus.
#pragma once
#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace povar_host {

// a device-resident Linearizor owns the authoritative state; BalProblem forwards its
// backup/restore calls (bal_bundle_adjustment.cpp:402, 508, 696, 808) through this hook
struct StateMirror {
  virtual ~StateMirror() = default;
  virtual void backup_pOSE() = 0;
  virtual void restore_pOSE() = 0;
  virtual void backup_joint() = 0;
  virtual void restore_joint() = 0;
  virtual void normalize_joint() = 0;                 // bal_bundle_adjustment.cpp:700-705
  virtual void pull_state() = 0;                      // device -> BalProblem
};

// The reference keeps a landmark's observations in a std::map<int, Observation> (bal_problem.hpp:226): five million tree
// nodes on venice.  Same interface where this code uses it (emplace with the inserted flag, ordered iteration, size),
// one contiguous sorted array per landmark.
class ObsMap {
 public:
  using value_type = std::pair<int, std::array<double, 2>>;
  using iterator = std::vector<value_type>::iterator;
  using const_iterator = std::vector<value_type>::const_iterator;
  std::pair<iterator, bool> emplace(int cam, const std::array<double, 2>& uv) {
    if (v_.empty() || v_.back().first < cam) {  // files list a landmark's cameras in ascending order almost always
      v_.emplace_back(cam, uv);
      return {v_.end() - 1, true};
    }
    auto it = std::lower_bound(v_.begin(), v_.end(), cam, [](const value_type& a, int c) { return a.first < c; });
    if (it != v_.end() && it->first == cam) return {it, false};
    it = v_.insert(it, value_type(cam, uv));
    return {it, true};
  }
  iterator begin() { return v_.begin(); }
  iterator end() { return v_.end(); }
  const_iterator begin() const { return v_.begin(); }
  const_iterator end() const { return v_.end(); }
  size_t size() const { return v_.size(); }
  bool empty() const { return v_.empty(); }
  void clear() { v_.clear(); }

 private:
  std::vector<value_type> v_;
};

class BalProblem {
 public:
  struct Camera {
    std::array<double, 12> space_matrix{};  // row-major 3x4 (bal_problem.hpp:102)
    std::array<double, 3> intrinsics{};     // f, k1, k2: parsed, carried, never used
    std::array<double, 12> space_matrix_backup{};
  };
  struct Landmark {
    std::array<double, 3> p_w{};
    std::array<double, 4> p_w_homogeneous{};
    ObsMap obs;  // camera index -> (u, v), v negated on load; ascending camera index like the reference's std::map
    std::array<double, 3> p_w_backup{};
    std::array<double, 4> p_w_homogeneous_backup{};
  };

  void load_bal_eccv(const std::string& path);                        // bal_problem.cpp:183-303
  void load_bal_varproj_space_matrix_write(const std::string& path, int seed);  // bal_problem.cpp:307-471

  std::vector<Camera>& cameras() { return cameras_; }
  std::vector<Landmark>& landmarks() { return landmarks_; }
  const std::vector<Camera>& cameras() const { return cameras_; }
  const std::vector<Landmark>& landmarks() const { return landmarks_; }
  int num_cameras() const { return (int)cameras_.size(); }
  int num_landmarks() const { return (int)landmarks_.size(); }
  long num_observations() const;

  void backup_pOSE();    // bal_problem.cpp:670-677
  void restore_pOSE();   // bal_problem.cpp:701-708
  void backup_joint();   // bal_problem.cpp:658-665
  void restore_joint();  // bal_problem.cpp:691-698

  // CSR export in the layout of include/povar_hip.h
  void flatten(std::vector<int>& lm_off, std::vector<int>& cam_idx, std::vector<double>& obs) const;

  StateMirror* mirror = nullptr;
  bool quiet = false;
  // A device-resident Linearizor leaves its context here when it is destroyed; the next one built on the same problem
  // with the same options (step 2 after step 1, bal_bundle_adjustment.cpp:283 / 585) takes it over instead of building
  // the device layout a second time.  key: what the context was built for (sizes, options, a hash of the observations).
  // The cache is not copied with the problem (a copy would share ONE live device context with the original: two
  // linearizors could adopt it under the same key) -- a copied problem starts without one; moves keep it.
  struct DeviceCache {
    std::shared_ptr<void> ctx;
    std::string key;
    DeviceCache() = default;
    DeviceCache(const DeviceCache&) {}
    DeviceCache& operator=(const DeviceCache& o) {
      if (this != &o) { ctx.reset(); key.clear(); }
      return *this;
    }
    DeviceCache(DeviceCache&&) = default;
    DeviceCache& operator=(DeviceCache&&) = default;
    void clear() { ctx.reset(); key.clear(); }
  };
  DeviceCache device_cache;

 private:
  std::vector<Camera> cameras_;
  std::vector<Landmark> landmarks_;
};

BalProblem load_normalized_bal_problem(const struct BalDatasetOptions& options, struct DatasetSummary* dataset_summary = nullptr,
                                       struct PipelineTimingSummary* timing_summary = nullptr);

}  // namespace povar_host
