/*
 * povar_hip.h -- C ABI of the MI355X (gfx950) implementation of PoVar's power-series
 * Schur-complement inner solve.
 *
 * The reference (tum-vision/povar, C++17, CPU only) has no C ABI.  Its operator boundary for
 * this path is the abstract class Linearizor<Scalar> (src/rootba_povar/solver/linearizor.hpp:48-82)
 * as implemented by LinearizorPowerVarproj (solver/linearizor_power_varproj.cpp:21-308) on top
 * of LinearizationPowerVarproj (sc/linearization_power_varproj.hpp:28-469), and -- for the explicit
 * Schur-complement solver types PCG / CHOLESKY / RIPCG -- by LinearizorSC (solver/linearizor_sc.cpp:50-360)
 * on top of LinearizationSC (sc/linearization_sc.hpp).  Each entry point below names the reference
 * interface it replaces.  A reference-side binding (a Linearizor
 * subclass forwarding to these calls) is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, caller-allocated HOST buffers, fp64 values, int32
 * indices, row-major 3x4 camera matrices flattened to 12 (bal_problem.hpp:147-157).  Every
 * function returns an int status: 0 = ok, POVAR_NUMERIC_FAILURE (1) = non-finite values where
 * the reference would flag numerical failure, < 0 = HIP / RCCL / argument error (text via
 * povar_last_error()).  No exceptions cross the ABI.  A context is single-threaded, like the
 * reference Linearizor (not re-entrant: linearizor_base.hpp:80-86).
 */
#ifndef POVAR_HIP_H
#define POVAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct povar_ctx povar_ctx;

#define POVAR_OK 0
#define POVAR_NUMERIC_FAILURE 1

/* BalResidualOptions::RobustNorm  (bal/bal_residual_options.hpp:45-49) */
enum { POVAR_NORM_NONE = 0, POVAR_NORM_HUBER = 1, POVAR_NORM_CAUCHY = 2 };
/* SolverOptions::SolverType values served by this path (bal/solver_options.hpp:60-66;
 * factory solver/linearizor.cpp:51-56) */
enum { POVAR_POWER_VARPROJ = 0, POVAR_POWER_SCHUR_COMPLEMENT = 1 };
/* LinearizationPowerVarproj::Summary::TerminationType (sc/linearization_power_varproj.hpp:52-56) */
enum { POVAR_LINEAR_SOLVER_NO_CONVERGENCE = 0, POVAR_LINEAR_SOLVER_SUCCESS = 1,
       POVAR_LINEAR_SOLVER_FAILURE = 2 };
/* how E0*x is evaluated: 0 = implicit (tiles recomputed from cameras/landmarks/observations,
 * deterministic two-pass scatter), 1 = stored tiles (the reference's [Jp|Jl] blocks kept in HBM
 * and streamed once per term, same deterministic scatter), 2 = implicit with the Jp^T s contributions
 * of the ~590 most observed cameras accumulated in LDS (fp64 LDS atomics: fastest, summation order
 * and therefore the last bits not reproducible run to run -- like the reference's mutex order),
 * 3 = stored tiles with the same LDS accumulation (Jp^T s taken from the stored Jp).  Step 2 has the
 * implicit forms only: modes 1 and 3 run its deterministic implicit path. */
enum { POVAR_E0_IMPLICIT = 0, POVAR_E0_TILES = 1, POVAR_E0_IMPLICIT_LDSACC = 2, POVAR_E0_TILES_LDSACC = 3 };

/* LandmarkBlockSC::Options (sc/landmark_block.hpp:61-75) + device selection */
typedef struct {
  int32_t robust_norm;       /* POVAR_NORM_* */
  double huber_parameter;    /* bal_residual_options.hpp:61-62 */
  double jacobi_scaling_eps; /* effective epsilon, linearizor_base.cpp:94-100 (1e-5 if option is 0) */
  int32_t device;            /* HIP device ordinal */
  int32_t e0_mode;           /* POVAR_E0_* */
  uint32_t flags;            /* POVAR_FLAG_* behaviour switches, 0 = the library's defaults (below) */
} povar_options;

/* povar_options.flags -- the behaviour switches of a context.  The reference passes every such switch through SolverOptions
 * (bal/solver_options.hpp:95-305); this is that struct's place for the MI355X build's own.  The POVAR_* environment variables
 * named beside them stay as OVERRIDES for diagnosis (a variable that is set wins over the flag). */
enum {
  /* run-to-run BIT-reproducible results for a given device count (fixed reduction order inside a GPU): the gather
   * mode for linearisation / preparation / cost, the fixed-point, ticket-ordered camera-chunk kernels for the terms (1.26 x the
   * default term on venice-1778), no row placement on a host thread, no timing decides a kernel.  [POVAR_DETERMINISTIC=1] */
  POVAR_FLAG_DETERMINISTIC = 1u << 0,
  /* with POVAR_FLAG_DETERMINISTIC: the terms in the gather form too (3.7 x the default term).  [POVAR_DET_CK=0] */
  POVAR_FLAG_DET_GATHER_TERMS = 1u << 1,
  /* launch the term loop kernel by kernel instead of replaying a captured hipGraph.  [POVAR_NO_GRAPH=1] */
  POVAR_FLAG_NO_GRAPH = 1u << 2,
  /* bits 4..7: per-term E0 kernel, as povar_set_e0_kernel: POVAR_FLAG_E0_KERNEL(k), k = -1 (field 0: the library times the
   * kernels once per layout and keeps the faster), 0 = e0_lpl / e0_lpl_h, 1..6 = a camera-chunk instantiation.  [POVAR_E0_CK=k] */
  POVAR_FLAG_E0_KERNEL_SHIFT = 4, POVAR_FLAG_E0_KERNEL_MASK = 0xFu << 4,
  /* bits 8..9: the m-term loop, as povar_set_series_kernel: POVAR_FLAG_SERIES_KERNEL(m), m = -1 (field 0: timed), 0 = per-term
   * kernels, 1 = the resident launch wherever the context allows it.  [POVAR_RES=m] */
  POVAR_FLAG_SERIES_KERNEL_SHIFT = 8, POVAR_FLAG_SERIES_KERNEL_MASK = 0x3u << 8,
  /* bits 12..13: LDS bank placement of the lane-per-landmark rows: POVAR_FLAG_PLACEMENT(p), p = 0 (by size: inside povar_create
   * under 2^20 observations, on a host thread from there on), 1 = inside povar_create, 2 = on a host thread, 3 = none.
   * [POVAR_LPL_PLACE=sync|async|none] */
  POVAR_FLAG_PLACEMENT_SHIFT = 12, POVAR_FLAG_PLACEMENT_MASK = 0x3u << 12,
  /* keep the 16-byte image points in the camera-chunk rows even where they pack (six-decimal observations).  [POVAR_CK_PACK=0] */
  POVAR_FLAG_NO_PACKED_ROWS = 1u << 16
};
#define POVAR_FLAG_E0_KERNEL(k) ((((uint32_t)((k) + 1)) & 0xFu) << POVAR_FLAG_E0_KERNEL_SHIFT)
#define POVAR_FLAG_SERIES_KERNEL(m) ((((uint32_t)((m) + 1)) & 0x3u) << POVAR_FLAG_SERIES_KERNEL_SHIFT)
#define POVAR_FLAG_PLACEMENT(p) ((((uint32_t)(p)) & 0x3u) << POVAR_FLAG_PLACEMENT_SHIFT)

/* ResidualInfo (bal/residual_info.hpp:59-92) */
typedef struct {
  int64_t all_num_obs;
  double all_error;
  double all_residual_sum;
  int64_t valid_num_obs;
  double valid_error;
  double valid_residual_sum;
  int32_t is_numerically_valid;
} povar_residual_info;

/* per-kernel-class device time measured with HIP events on the context's stream */
typedef struct {
  double e0_ms;      /* sum over launches of the E0 (SpMV) kernels of one term */
  int64_t e0_launches;
  double binv_ms;    /* B^-1 / AXPY kernel */
  int64_t binv_launches;
  double comm_ms;    /* RCCL all-reduce */
  int64_t comm_launches;
} povar_profile_info;

const char* povar_last_error(void);
/* number of visible HIP devices (0 without a GPU; < 0 on a runtime error) */
int povar_device_count(void);
/* compute units of a device (hipDeviceProp_t::multiProcessorCount; < 0 on a runtime error).  A launcher that puts several
 * ranks on ONE device gives each a range of them (environment POVAR_CU_MASK=<first>-<last>, read by povar_create). */
int povar_device_cu_count(int32_t device);

/* LinearizorPowerVarproj ctor (linearizor_power_varproj.cpp:21-38) + LinearizationVarProj ctor /
 * allocate_landmark (linearization_varproj.hpp:42-61, landmark_block.hpp:101-133).
 * lm_offsets[n_lms+1] CSR over observations, cam_idx ascending inside a landmark (std::map
 * order, bal_problem.hpp:226), obs = (u, v) with v already negated (bal_problem.cpp:240).
 * With povar_comm_init() the arrays describe this rank's landmark shard only. */
int povar_create(povar_ctx** out, int32_t n_cams, int32_t n_lms, int64_t n_obs,
                 const int32_t* lm_offsets, const int32_t* cam_idx, const double* obs,
                 const povar_options* options);
void povar_destroy(povar_ctx* ctx);

/* BalProblem state shared with the caller (linearizor_base.hpp:80; bal_problem.hpp:290-294) */
int povar_set_cameras(povar_ctx* ctx, const double* cams /*12 n_cams*/);
int povar_get_cameras(povar_ctx* ctx, double* cams);
int povar_set_landmarks(povar_ctx* ctx, const double* lms /*3 n_lms*/);
int povar_get_landmarks(povar_ctx* ctx, double* lms);
/* BalProblem::backup_pOSE / restore_pOSE (bal_problem.cpp:670-677, 701-708) */
int povar_backup_pose(povar_ctx* ctx);
int povar_restore_pose(povar_ctx* ctx);

/* Linearizor::initialize_varproj_lm_pOSE (linearizor_base.cpp:60-67; helper.cpp:76-114) */
int povar_init_landmarks_pose(povar_ctx* ctx, double alpha);
/* Linearizor::compute_error_pOSE (linearizor_base.cpp:69-77; helper.cpp:117-154) */
int povar_error_pose(povar_ctx* ctx, double alpha, povar_residual_info* out);
/* Linearizor::linearize_pOSE (linearizor_power_varproj.cpp:45-76) */
int povar_linearize_pose(povar_ctx* ctx, double alpha);
/* Linearizor::solve (linearizor_power_varproj.cpp:178-243): scale_Jp_cols on a new
 * linearisation point, prepare_Hb_pOSE[_poBA], solve_pOSE.  inc[12 n_cams] in scaled
 * coordinates; *num_iterations / *termination as Summary (linearization_power_varproj.hpp:51-61).
 * q_tolerance = SolverOptions::eta, r_tolerance = SolverOptions::r_tolerance. */
int povar_solve_pose(povar_ctx* ctx, double lambda, int32_t solver_type, int32_t power_sc_iterations,
                     double q_tolerance, double r_tolerance, double* inc, int32_t* num_iterations,
                     int32_t* termination);
/* Linearizor::apply (linearizor_power_varproj.cpp:246-273): camera update + back substitution;
 * *l_diff = model cost change. */
int povar_apply_pose(povar_ctx* ctx, int32_t solver_type, double alpha, const double* inc,
                     double* l_diff);

/* ---- the two halves of povar_solve_pose, separately (bench / parity tests) ---- */
/* LinearizationPowerVarproj::prepare_Hb_pOSE / _poBA (linearization_power_varproj.hpp:124-188) */
int povar_prepare_pose(povar_ctx* ctx, double lambda, int32_t solver_type);
/* LinearizationPowerVarproj::solve_pOSE (linearization_power_varproj.hpp:191-237); asynchronous
 * when both tolerances are <= 0 (result stays on the device; see povar_get_increment). */
int povar_power_series_pose(povar_ctx* ctx, int32_t power_sc_iterations, double q_tolerance,
                            double r_tolerance, int32_t* num_iterations, int32_t* termination);
int povar_get_increment(povar_ctx* ctx, double* inc /*12 n_cams*/);
/* term-by-term access: begin = accum = tmp = B^-1(-b) (hpp:196-200), step = one pass of the loop
 * body (hpp:202-203); get_term copies the current term. */
int povar_power_series_begin(povar_ctx* ctx);
int povar_power_series_step(povar_ctx* ctx);
int povar_get_term(povar_ctx* ctx, double* term /*12 n_cams*/);
/* LinearizationPowerVarproj::right_mul_e0_pOSE (linearization_power_varproj.hpp:364-406) */
int povar_right_mul_e0_pose(povar_ctx* ctx, const double* x, double* y);
int povar_set_e0_mode(povar_ctx* ctx, int32_t e0_mode);
int povar_synchronize(povar_ctx* ctx);

/* ---- step 2: projective refinement on the Riemannian manifold (RIPOBA) ---- */
/* BalProblem::Landmark::p_w_homogeneous (bal_problem.hpp:225), 4 n_lms; shares storage with p_w */
int povar_set_landmarks_homogeneous(povar_ctx* ctx, const double* lms_h);
int povar_get_landmarks_homogeneous(povar_ctx* ctx, double* lms_h);
/* BalProblem::backup_joint / restore_joint (bal_problem.cpp:658-665, 691-698) */
int povar_backup_joint(povar_ctx* ctx);
int povar_restore_joint(povar_ctx* ctx);
/* Linearizor::compute_error_homogeneous (linearizor_base.cpp:79-87; helper.cpp:157-196) */
int povar_error_homogeneous(povar_ctx* ctx, povar_residual_info* out);
/* Linearizor::linearize_projective_space_homogeneous (linearizor_power_varproj.cpp:80-110) */
int povar_linearize_homogeneous(povar_ctx* ctx);
/* LinearizationPowerVarproj::prepare_Hb_joint (linearization_power_varproj.hpp:74-122) incl. the
 * scale_Jp_cols_joint + linearize_nullspace of a new linearisation point (cpp:129-133) */
int povar_prepare_joint(povar_ctx* ctx, double lambda);
/* Linearizor::solve_joint (linearizor_power_varproj.cpp:114-175); inc[11 n_cams] in tangent
 * coordinates of the Householder bases N_c (see povar_get_buffer POVAR_BUF_NC_HOUSEHOLDER) */
int povar_solve_joint(povar_ctx* ctx, double lambda, int32_t power_sc_iterations, double q_tolerance,
                      double r_tolerance, double* inc, int32_t* num_iterations, int32_t* termination);
/* Linearizor::apply_joint (linearizor_power_varproj.cpp:277-308) */
int povar_apply_joint(povar_ctx* ctx, const double* inc, double* l_diff);
/* the renormalisation the outer loop performs after apply_joint (bal_bundle_adjustment.cpp:700-705) */
int povar_normalize_joint(povar_ctx* ctx);
/* povar_power_series_pose / _begin / _step / povar_get_term / povar_get_increment act on the system
 * prepared last (povar_prepare_pose: 12 n_cams vectors, povar_prepare_joint: 11 n_cams vectors) */

/* ---- explicit-Schur-complement solvers: LinearizorSC (solver/linearizor_sc.cpp), selected by
 * --solver-type-step-1 PCG | CHOLESKY and --solver-type-step-2 RIPCG (solver/linearizor.cpp:51-53,
 * 67-72).  The reduced camera system S x = b, S = (Hpp + lambda I) - E0, is solved without landmark
 * damping in step 1 (linearizor_sc.cpp:85-160) and with it in step 2 (:224-303); linearisation,
 * error evaluation and apply are the calls above (povar_apply_pose with POVAR_POWER_VARPROJ ==
 * LinearizorSC::apply, linearizor_sc.cpp:65-83). ---- */
enum { POVAR_SC_PCG = 0, POVAR_SC_CHOLESKY = 1 };
/* LinearizorPowerVarproj::linearize_pOSE scales the Jl columns (linearizor_power_varproj.cpp:62-64 ->
 * landmark_block.hpp:284-295), LinearizorSC::linearize_pOSE does not (linearizor_sc.cpp:163-191); the
 * stored Jl enters l_diff of apply (landmark_block.hpp:703-704).  enable = 1 (default) / 0 takes effect
 * at the next povar_linearize_pose.  Step 2 scales in both linearizors. */
int povar_set_jl_col_scaling(povar_ctx* ctx, int32_t enable);
/* LinearizorSC::solve (linearizor_sc.cpp:85-160).  POVAR_SC_PCG: ConjugateGradientsSolver::solve
 * (cg/conjugate_gradient.hpp:112-290) as driven by LinearizorBase::pcg (linearizor_base.cpp:104-125:
 * min/max_linear_solver_iterations, q_tolerance = eta, r_tolerance = -1, result negated) with the
 * SCHUR_JACOBI preconditioner (cg/preconditioner.hpp:66-135); S is applied matrix-free with the
 * per-term E0 kernels.  POVAR_SC_CHOLESKY: solve_direct_pOSE (sc/linearization_sc.hpp:236-245), dense
 * Cholesky of S on the device (needs 8 * (12 n_cams)^2 bytes of HBM; num_iterations = 0).
 * termination = POVAR_LINEAR_SOLVER_*. */
int povar_solve_pose_sc(povar_ctx* ctx, double lambda, int32_t method, int32_t min_iterations,
                        int32_t max_iterations, double eta, double* inc, int32_t* num_iterations,
                        int32_t* termination);
/* LinearizorSC::solve_joint (linearizor_sc.cpp:224-303): PCG (conjugate_gradient.hpp:292-470) on the
 * 11 n_cams tangent system */
int povar_solve_joint_sc(povar_ctx* ctx, double lambda, int32_t min_iterations, int32_t max_iterations,
                         double eta, double* inc, int32_t* num_iterations, int32_t* termination);

/* ---- inspection of internal state in the reference's layouts (parity tests) ---- */
enum {
  POVAR_BUF_DIAG2 = 0,     /* get_Jp_diag2_pOSE, 12 n_cams (linearization_varproj.hpp:183-222) */
  POVAR_BUF_POSE_SCALING,  /* pose_jacobian_scaling_pOSE_, 12 n_cams (linearizor_power_varproj.cpp:68-70) */
  POVAR_BUF_JL_COL_SCALE,  /* Jl_col_scale_pOSE, 3 n_lms (landmark_block.hpp:289-292) */
  POVAR_BUF_HLL_INV,       /* hll_inv_pOSE_, 9 n_lms (linearization_power_varproj.hpp:469) */
  POVAR_BUF_B,             /* b_p, 12 n_cams */
  POVAR_BUF_B_INV,         /* b_inv_pOSE_, 144 n_cams */
  POVAR_BUF_STORAGE,       /* storage_pOSE_ of every landmark, [4 n_obs][16] (landmark_block.hpp:726) */
  POVAR_BUF_JL_COL_SCALE_H, /* Jl_col_scale_homogeneous, 4 n_lms (landmark_block.hpp:303-306) */
  POVAR_BUF_B_JOINT,       /* b_p of prepare_Hb_joint, 11 n_cams */
  POVAR_BUF_B_INV_JOINT,   /* b_inv_joint_, 121 n_cams */
  POVAR_BUF_NC_HOUSEHOLDER, /* per camera (w[12], beta): N_c = (I - beta w w^T)[:, 1:], 13 n_cams */
  POVAR_BUF_SC_PRECOND,    /* Schur-Jacobi inv_blocks of the last povar_solve_*_sc (preconditioner.hpp:80-107),
                              144 n_cams (step 1) or 121 n_cams (step 2) */
  POVAR_BUF_SC_BLOCKDIAG   /* B_c = Hpp_c + lambda I of the last povar_solve_*_sc, same sizes */
};
int povar_get_buffer(povar_ctx* ctx, int32_t which, double* out, int64_t n);

/* ---- measurement ---- */
int povar_profile_enable(povar_ctx* ctx, int32_t enable);
int povar_profile_get(povar_ctx* ctx, povar_profile_info* out);
/* bytes of device memory held by the context */
int64_t povar_device_bytes(povar_ctx* ctx);
/* Byte floor of ONE application of E0 (right_mul_e0_pOSE, linearization_power_varproj.hpp:364-406) in the
 * context's current E0 mode: what the landmark-major kernel and the per-camera kernel must stream by design
 * (each array once).  bench.py prices the HIP-event time of those kernels against it (roofline.achieved). */
int povar_e0_model_bytes(povar_ctx* ctx, int64_t* lm_kernel_bytes, int64_t* cam_kernel_bytes);
/* Device-side stage timings (hipEvent pairs on the context's stream around the entry points), the source of the
 * IterationSummary timing fields of solver/solver_summary.hpp:186-212 that LinearizorPowerVarproj fills from host
 * timers (linearizor_power_varproj.cpp:61-72, 194-233, 257-258).  Accumulated since povar_timings_enable(ctx, 1). */
typedef struct {
  double linearize_ms;  /* linearize_pOSE / linearize_projective_space_homogeneous: stage1_time */
  int64_t linearize_calls;
  double prepare_ms;    /* prepare_Hb_*: prepare_time (+ scale_pose_jacobian, landmark_damping) */
  int64_t prepare_calls;
  double solve_ms;      /* solve_pOSE / solve_joint (power series) or PCG / CHOLESKY: solve_reduced_system_time */
  int64_t solve_calls;
  double apply_ms;      /* apply / apply_joint: back_substitution_time + update_cameras_time */
  int64_t apply_calls;
  double other_ms;      /* cost evaluations, landmark initialisation */
  int64_t other_calls;
} povar_timings_info;
int povar_timings_enable(povar_ctx* ctx, int32_t enable);
int povar_timings(povar_ctx* ctx, povar_timings_info* out);
/* what the layout of the per-term E0 kernel decided for this problem (measurement / DESIGN.md tables) */
typedef struct {
  int32_t grid;         /* E0 workgroups */
  int32_t lds_slots;    /* camera slots (record + accumulators) of the fullest workgroup */
  int32_t n_global;     /* cameras resident in every workgroup */
  int32_t n_tail;       /* cameras resident in one row and one column of the workgroup grid */
  int64_t n_tiles, n_rows;  /* 64-lane tiles and observation rows of the row stream */
  int64_t n_cold;       /* observations whose camera is not LDS-resident in their workgroup */
  int64_t n_obs;
  int64_t lane_per_landmark; /* 1: the term loop runs e0_lpl / e0_lpl_h on this layout; 0: the lane-per-observation
                                kernels (problems under 65 536 observations, POVAR_E0_V1=1) */
  double create_ms;     /* host wall time of povar_create: layout construction + uploads (the reference's counterpart,
                           allocating the landmark blocks, sc/linearization_varproj.hpp:44-60, is part of its
                           preprocessor_time_in_seconds too, bal_bundle_adjustment.cpp:260-286) */
  int32_t strategy;     /* landmark -> workgroup assignment the layout chose: 0 rank-based camera grid (graphs without
                           locality), 1 contiguous landmark ranges with per-workgroup camera sets (the file's landmark
                           order carries the locality, as for the reference: bal/bal_problem.cpp:183-303) */
  int32_t hubs;         /* camera slots per workgroup with four accumulator replicas */
  int32_t placement;    /* LDS bank placement of the rows: 0 natural order (none), 1 placed inside povar_create,
                           2 being placed on a host thread (the kernels run on the natural order meanwhile),
                           3 placed rows swapped in */
  double placement_ms;  /* host wall time of the background placement (0 until it has finished) */
  int32_t e0_kernel;    /* per-term E0 kernel of step 1: 0 e0_lpl (lane = landmark, cameras in LDS), 1..6: the e0_ck
                           instantiation in use (lane = camera chunk, landmarks in LDS; povar_set_e0_kernel); with
                           POVAR_DETERMINISTIC=1: 7 = e0_ck_det (the bit-reproducible camera-chunk kernel), 0 = the gather form */
  int32_t ck_ready;     /* 1: the camera-chunk layout of the rows in use exists (with the rows placed on a host thread
                           it arrives together with them) */
  int32_t ck_batches, ck_slots;  /* landmark batches per workgroup, landmark slots per batch */
  int32_t ck_tiles_max; /* most chunk tiles of one (workgroup, batch) */
  int32_t ck_part_rec;  /* partial records of the camera-chunk kernel (workgroup slots + chunks of cameras without one) */
  int64_t ck_rows, ck_chunks, ck_cold_chunks;
  double ck_build_ms;   /* host time of the derivation from the lane-per-landmark layout */
  int32_t e0_auto;      /* 0: the E0 kernel was forced (POVAR_E0_CK, povar_set_e0_kernel); 1: the library will time e0_lpl and e0_ck
                           on this problem at the next power series; 2: it has (e0_kernel is its choice) */
  float tune_lpl_us, tune_ck_us;  /* what that timing saw: microseconds per term pair (the E0 kernel + the per-camera kernel behind it), the faster of two rounds */
  /* step 2 (solve_joint): the same pair of kernels for the homogeneous operator, on a layout instance of its own
   * (64 instead of 48 bytes of LDS per landmark slot: more batches, shorter chunks) */
  int32_t e0_kernel_h;  /* 0: e0_lpl_h, 1: e0_ck_h; with POVAR_DETERMINISTIC=1: 2 = e0_ck_h_det, 0 = the gather form */
  int32_t ckh_ready, ckh_batches, ckh_slots;
  int64_t ckh_chunks, ckh_cold_chunks;
  int32_t e0_auto_h;    /* as e0_auto: 0 forced, 1 to be timed at the next step-2 power series, 2 timed */
  float tune_lpl_h_us, tune_ck_h_us;
  /* resident power series (series_res, povar_kernels_res.hpp): the whole loop of solve_pOSE
   * (sc/linearization_power_varproj.hpp:191-237) in one launch, observation rows in registers, landmarks and B^-1 in LDS */
  int32_t res_ready;    /* 1: the layout exists (the context's observations fit the lanes of one workgroup per CU) */
  int32_t res_active;   /* 1: the next step-1 power series of this context runs as the resident kernel */
  int32_t res_auto;     /* 0 forced (POVAR_RES, povar_set_series_kernel), 1 to be timed at the next series, 2 timed */
  int32_t res_wgs, res_waves, res_rows, res_rounds;  /* workgroups, wavefronts per workgroup, rows per chunk, chunks per lane */
  int32_t res_records;  /* partial records = (workgroup, camera) pairs */
  int32_t res_max_cams, res_max_lms, res_max_chunks, res_max_oq;  /* of the fullest workgroup: cameras, landmarks, chunks,
                           partial records it reads as the owner of cameras */
  int32_t res_order;    /* landmark order the workgroup ranges were cut from: 0 natural (file) order, 1 by rarest camera */
  int32_t res_lds_bytes;
  double res_build_ms;
  float tune_terms_us, tune_res_us;  /* what the timing saw, microseconds per term: per-term kernels, resident series */
  int32_t res_failed;   /* 1: a resident series gave up (its workgroups were not all on the device together); the
                           context repeated that series with the per-term kernels and stays on them */
  int32_t ck_packed;    /* 1: the rows of the camera-chunk layout keep the image points packed (two int32 of micro-units,
                           8 instead of 16 bytes per observation and walk): every observation of the problem is a six-decimal
                           number -- what the reference's files hold, bal/bal_problem.cpp:373-375 -- and comes back bit for bit */
  int32_t ck_cold_q;    /* 1: chunks of cameras without an accumulator slot leave q of each observation (32 bytes) in the cold
                           camera-major view instead of a 96-byte partial record per chunk (layouts with at most 8 % cold observations) */
  int32_t ckh_stride;   /* step 2's camera-chunk layout: landmark slots the LDS arrays of e0_ck_h are cut for -- 1536, or 2048 where
                           that saves a landmark batch (then with ckh_accumulators < lds_slots: the workgroup keeps the accumulator
                           slots of its most observed cameras, the other cameras' chunks write records of their own) */
  int32_t ckh_accumulators;      /* accumulator slots per workgroup at most */
  int64_t ckh_capped_obs;        /* observations of cameras that have a slot in the lane-per-landmark layout but none here */
} povar_layout_info;
int povar_get_layout_info(povar_ctx* ctx, povar_layout_info* out);
/* The reference's constructor is a trivial allocation (sc/linearization_varproj.hpp:44-60); this library's builds the
 * lane-per-landmark row stream, and the LDS bank placement of its rows is two thirds of that time while it only buys
 * 10 % of the term rate.  From 2^20 observations on (POVAR_LPL_PLACE=sync|async|none overrides) povar_create therefore
 * returns on the natural row order and a host thread places the rows; they are swapped in by the first
 * povar_linearize_* call that finds them ready.  povar_layout_finalize(ctx, 1) waits for the thread and swaps at
 * once (benchmarks; a linearisation taken before is dropped: linearise again), (ctx, 0) swaps only if ready.
 * Returns 1 if the placed rows are in use after the call, 0 if not (yet), < 0 on error. */
int povar_layout_finalize(povar_ctx* ctx, int32_t wait);
/* Per-term E0 kernels (step 1 replaces right_mul_e0_pOSE, sc/linearization_power_varproj.hpp:364-406; step 2
 * right_mul_e0_joint, :408-453; either way): 0 = e0_lpl / e0_lpl_h; > 0 = the camera-chunk kernels: for step 2 e0_ck_h
 * (povar_kernels_ck_joint.hpp), for step 1 one of the e0_ck instantiations (povar_kernels_ck.hpp; table POVAR_CK_VARIANTS in povar_ctx.hpp: wavefronts per
 * workgroup, register-resident tiles, rows in flight).  Environment: POVAR_E0_CK=<n> sets the initial choice, which also
 * decides how the camera-chunk layout is cut (chunk cap, wavefronts the tiles are scheduled over).  kernel = -1 (the
 * default when nothing is forced): the library times both kernels of a step once per layout on the prepared problem and
 * keeps the faster one (povar_layout_info.e0_auto[_h], tune_*_us). */
int povar_set_e0_kernel(povar_ctx* ctx, int32_t kernel);
/* The m-term loop of solve_pOSE (sc/linearization_power_varproj.hpp:191-237) as per-term kernels inside a hipGraph
 * (mode 0) or as ONE resident launch that keeps the term-invariant operands on the chip (mode 1; contexts of up to 400 000
 * observations -- POVAR_RES_MAX_OBS -- in the LDS-accumulating E0 mode, without peers: a context with a communicator of more
 * than one rank or with the peer-to-peer exchange runs the per-term kernels);
 * -1 (default): the library times both once per context on the caller's prepared system and keeps the faster one
 * (povar_layout_info.res_auto, tune_terms_us, tune_res_us).  Environment: POVAR_RES=0|1. */
int povar_set_series_kernel(povar_ctx* ctx, int32_t mode);
/* Diagnostic builds only (tools/variants/build_variant.sh ck_stamps -- a patched copy of the sources --, tools/ck_stamps.py):
 * in-kernel s_memtime stamps of e0_ck's phases, [workgroups][16 wavefronts][40]; the first call arms the collection.  The
 * shipped library executes no stamp and returns an error. */
int povar_debug_ck_stamps(povar_ctx* ctx, uint64_t* out, int64_t n);

/* ---- multi-GPU: landmarks sharded over ranks, one RCCL all-reduce per exchange step ---- */
/* host-only: contiguous landmark range of `rank`, balanced by observation count */
int povar_shard_range(int32_t n_lms, const int32_t* lm_offsets, int32_t world, int32_t rank,
                      int32_t* lm_begin, int32_t* lm_end);
int povar_comm_unique_id(uint8_t id[128]);
int povar_comm_init(povar_ctx* ctx, int32_t world, int32_t rank, const uint8_t id[128]);
/* ranks of the attached communicator (ncclCommCount; the world size of a host hook), 0 = none, < 0 on error */
int povar_comm_ranks(povar_ctx* ctx);
/* Peer-to-peer exchange of the per-term E0 partials, on top of a communicator attached with
 * povar_comm_init / povar_comm_init_host (which keeps serving the once-per-solve exchanges): instead of one
 * all-reduce of 12 n_cams doubles per power-series term, every rank's per-camera kernel pushes its partial sums
 * into every peer's exchange buffer (stores over xGMI) and the B^-1 kernel of every rank waits for the world's slabs
 * of its camera and sums them in rank order -- two kernels and no library call per term, deterministic for a world.
 * Every rank calls povar_p2p_export(ctx, world, handle) (allocates its buffer, returns a 64-byte hipIpcMemHandle),
 * the launcher gathers the handles, every rank calls povar_p2p_attach(ctx, world, rank, handles[world][64]). */
int povar_p2p_export(povar_ctx* ctx, int32_t world, uint8_t handle[64]);
int povar_p2p_attach(povar_ctx* ctx, int32_t world, int32_t rank, const uint8_t* handles);
/* switch the per-term exchange between the attached peer-to-peer kernels (on != 0) and the communicator's
 * all-reduce (0): a launcher validates the first against the second before it trusts it (bench.py does) */
int povar_p2p_enable(povar_ctx* ctx, int32_t on);
/* same exchange steps through a caller-supplied host all-reduce (sum, in place) instead of RCCL:
 * lets an MPI/gloo launcher or an in-process test stand in for the communicator */
typedef void (*povar_allreduce_fn)(double* buf, int64_t n, void* user);
int povar_comm_init_host(povar_ctx* ctx, int32_t world, int32_t rank, povar_allreduce_fn fn, void* user);

#ifdef __cplusplus
}
#endif
#endif
