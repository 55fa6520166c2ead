/*
 * povar_oracle.h -- CPU restatement of PoVar's power-series Schur-complement path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference (tum-vision/povar) ships no
 * tests, golden vectors or fixtures for this path and cannot be built in this
 * image (Eigen 3.4 / Sophus / TBB headers / glog are absent, bin/bal is a macOS
 * Mach-O binary).  What pins this restatement instead: an independent NumPy
 * restatement (oracle/povar_numpy.py, dense linear algebra) whose outputs are
 * committed under tests/golden/, central-difference Jacobian checks, and the
 * dense Schur-complement limit of the power series (tests/test_oracle_*.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/rootba_povar/).  Third-party arithmetic the reference
 * delegates to Eigen 3.4.0 (un-vendored submodule external/eigen) is restated
 * from the published algorithm: fixed-size 3x3 inverse = cofactor/determinant
 * (Eigen/src/LU/InverseImpl.h, compute_inverse_size3_helper), LLT = Cholesky
 * using the upper triangle, colwise().norm() = sqrt of the in-order sum of
 * squares, bdcSvd().solve() = the least-squares solution (restated here with a
 * Householder QR, equal for full-column-rank G).
 *
 * Layouts (all fp64, indices int32):
 *   cams   [n_cams][12]   row-major 3x4 "space_matrix"           (bal_problem.hpp:102)
 *   lms    [n_lms][3]     p_w                                     (bal_problem.hpp:224)
 *   lms_h  [n_lms][4]     p_w_homogeneous                         (bal_problem.hpp:225)
 *   obs    [n_obs][2]     (u, v) with v already negated on load   (bal_problem.cpp:240)
 *   lm_off [n_lms+1]      CSR: observations of landmark l are [lm_off[l], lm_off[l+1])
 *   cam_idx[n_obs]        camera of each observation, ascending inside a landmark
 *                         (std::map order, bal_problem.hpp:226; landmark_block.hpp:104-108)
 *   storage[4*n_obs][16]  step 1: per landmark row-major [4k x 16] = [Jp(12) | Jl(3) | r(1)]
 *                         (landmark_block.hpp:120-126, 167-169), landmarks concatenated
 *   storage_h[2*n_obs][17] step 2 homogeneous: [Jp(12) | Jl(4) | r(1)] (landmark_block.hpp:112-114)
 *   storage_n[2*n_obs][14] step 2 tangent:     [Jp(11) | Jl(3)]         (landmark_block.hpp:116-118)
 */
#ifndef POVAR_ORACLE_H
#define POVAR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_NORM_NONE = 0, ORC_NORM_HUBER = 1, ORC_NORM_CAUCHY = 2 };
enum { ORC_POWER_VARPROJ = 0, ORC_POWER_SCHUR_COMPLEMENT = 1 };
enum { ORC_NO_CONVERGENCE = 0, ORC_SUCCESS = 1, ORC_FAILURE = 2 };

typedef struct {
  int32_t n_cams;
  int32_t n_lms;
  int64_t n_obs;
  const int32_t* lm_off;
  const int32_t* cam_idx;
  const double* obs;
} orc_problem;

typedef struct {
  int32_t robust_norm;     /* ORC_NORM_* (bal_residual_options.hpp:45-49) */
  double huber_parameter;  /* bal_residual_options.hpp:61-62 */
  double jacobi_scaling_eps; /* effective eps, linearizor_base.cpp:94-100 (1e-5 when the option is 0) */
} orc_options;

/* residual_info.hpp:59-92 */
typedef struct {
  int64_t all_num_obs;
  double all_error;
  double all_residual_sum;
  int64_t valid_num_obs;
  double valid_error;
  double valid_residual_sum;
  int32_t is_numerically_valid;
} orc_residual_info;

/* ---- L0: per-observation math (bal/bal_bundle_adjustment_helper.cpp) ---- */
void orc_error_weight(const orc_options* o, double res_squared, double* error, double* weight);
void orc_linearize_point_pose(double alpha, const double* obs, const double* x, const double* P,
                              double* res, double* Jp, double* Jl);
int orc_linearize_point_homogeneous(const double* obs, const double* X, const double* P,
                                    double* res, double* Jp, double* Jl);

/* ---- step 1 (pOSE / VarPro) ---- */
void orc_init_landmarks_pose(const orc_problem* p, double alpha, const double* cams, double* lms);
void orc_error_pose(const orc_problem* p, const orc_options* o, double alpha, const double* cams,
                    const double* lms, orc_residual_info* out);
int orc_linearize_pose(const orc_problem* p, const orc_options* o, double alpha, const double* cams,
                       const double* lms, double* storage);
void orc_jp_diag2_pose(const orc_problem* p, const double* storage, double* diag2);
void orc_scale_jl_cols_pose(const orc_problem* p, const orc_options* o, double* storage,
                            double* jl_col_scale);
void orc_scale_jp_cols_pose(const orc_problem* p, double* storage, const double* scaling);
void orc_prepare_hb_pose(const orc_problem* p, const double* storage, double lambda_pose,
                         double lambda_lm, double* hll_inv, double* b, double* b_inv);
void orc_right_mul_b_inv(int32_t n_cams, int32_t dim, const double* b_inv, const double* x, double* y);
void orc_right_mul_e0_pose(const orc_problem* p, const double* storage, const double* hll_inv,
                           const double* x, double* y);
/* reference-faithful threaded variant: tbb::parallel_for over landmarks (LPV:402-403) + per-camera mutex.
 * orc_set_e0_schedule(0): one contiguous landmark range per thread, balanced by observations (static partitioner);
 * orc_set_e0_schedule(g > 0): chunks of g landmarks taken on demand (TBB's default auto_partitioner analogue). */
void orc_set_e0_schedule(int32_t grain);
/* 0: per-camera mutex (the reference, LPV:393-397); 1: per-thread private sums joined afterwards (a second CPU baseline) */
void orc_set_e0_scatter(int32_t private_sums);
void orc_right_mul_e0_pose_mt(const orc_problem* p, const double* storage, const double* hll_inv,
                              const double* x, double* y, int32_t n_threads);
int orc_solve_pose(const orc_problem* p, const double* storage, const double* hll_inv,
                   const double* b_inv, const double* b, int32_t m, double q_tol, double r_tol,
                   double* accum, int32_t* num_iterations, double* terms, int32_t n_threads);
double orc_back_substitute_pose(const orc_problem* p, double alpha, const double* storage,
                                const double* cams, double* lms, const double* inc);
double orc_back_substitute_poba(const orc_problem* p, const double* storage,
                                const double* jl_col_scale, double lambda_lm, double* lms,
                                const double* inc);
void orc_apply_cam_inc(int32_t n_cams, double* cams, const double* inc);

/* ---- step 2 (projective refinement on the Riemannian manifold) ---- */
void orc_kernel_basis(int32_t n, const double* v, double* N);
void orc_error_homogeneous(const orc_problem* p, const orc_options* o, const double* cams,
                           const double* lms_h, orc_residual_info* out);
int orc_linearize_homogeneous(const orc_problem* p, const orc_options* o, const double* cams,
                              const double* lms_h, double* storage_h);
void orc_jp_diag2_homogeneous(const orc_problem* p, const double* storage_h, double* diag2);
void orc_scale_jl_cols_homogeneous(const orc_problem* p, const orc_options* o, double* storage_h,
                                   double* jl_col_scale_h);
void orc_scale_jp_cols_joint(const orc_problem* p, double* storage_h, const double* scaling);
void orc_linearize_nullspace(const orc_problem* p, const double* cams, const double* lms_h,
                             const double* storage_h, double* storage_n);
void orc_prepare_hb_joint(const orc_problem* p, const double* storage_h, const double* storage_n,
                          double lambda, double* hll_inv, double* b, double* b_inv);
void orc_right_mul_e0_joint(const orc_problem* p, const double* storage_n, const double* hll_inv,
                            const double* x, double* y);
int orc_solve_joint(const orc_problem* p, const double* storage_n, const double* hll_inv,
                    const double* b_inv, const double* b, int32_t m, double q_tol, double r_tol,
                    double* accum, int32_t* num_iterations, double* terms);
double orc_back_substitute_joint(const orc_problem* p, const double* storage_h,
                                 const double* jl_col_scale_h, double lambda, const double* cams,
                                 double* lms_h, const double* inc);
void orc_apply_cam_inc_joint(int32_t n_cams, double* cams, const double* inc11,
                             const double* scaling);
void orc_normalize_joint(int32_t n_cams, int32_t n_lms, double* cams, double* lms_h);

/* ---- explicit Schur complement (LinearizorSC: PCG / CHOLESKY / RIPCG), dense S [n][n] ---- */
void orc_get_hb_pose(const orc_problem* p, const double* storage, double lambda_pose, double* S,
                     double* b);
void orc_get_hb_joint(const orc_problem* p, const double* storage_h, const double* storage_n,
                      double lambda, double* S, double* b);
void orc_block_jacobi_inverse(int32_t n_cams, int32_t dim, const double* S, double* inv_blocks);
int orc_pcg(int32_t n_cams, int32_t dim, const double* S, const double* b, const double* inv_blocks,
            int32_t min_iterations, int32_t max_iterations, double eta, double* x,
            int32_t* num_iterations);
int orc_cholesky_solve(int32_t n, const double* S, const double* b, double* x);

#ifdef __cplusplus
}
#endif
#endif
