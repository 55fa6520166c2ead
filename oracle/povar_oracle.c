/*
 * povar_oracle.c -- CPU restatement of PoVar's power-series Schur-complement path.
 * TEST INFRASTRUCTURE ONLY (see povar_oracle.h).  "parity unpinned" by the reference;
 * pinned by oracle/povar_numpy.py goldens + self-consistency tests.
 *
 * Reference aliases (relative to /root/reference/src/rootba_povar/):
 *   HLP = bal/bal_bundle_adjustment_helper.cpp      LMB = sc/landmark_block.hpp
 *   LVP = sc/linearization_varproj.hpp              LPV = sc/linearization_power_varproj.hpp
 *   LZR = solver/linearizor_power_varproj.cpp       BBA = solver/bal_bundle_adjustment.cpp
 */
#define _GNU_SOURCE /* pthread_barrier_t under -std=c11 */
#include "povar_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ST 16  /* columns of storage_pOSE_: Jp(12) | Jl(3) | r(1)   LMB:120-123 */
#define SH 17  /* columns of storage_homogeneous_: Jp(12) | Jl(4) | r(1)   LMB:112-114 */
#define SN 14  /* columns of storage_nullspace_: Jp(11) | Jl(3)   LMB:116-117 */

/* ------------------------------------------------------------------------- */
/* small dense helpers (restating the Eigen 3.4.0 calls on the path)          */
/* ------------------------------------------------------------------------- */

/* Eigen fixed 3x3 .inverse(): cofactors / determinant (LMB:518, 551, 486, 702, 647, 614) */
static void inv3(const double* m, double* r) {
  /* cofactors_col0(i) = cofactor_3x3<i,0>; det = (cofactors_col0 .* matrix.col(0)).sum() */
  const double k00 = m[4] * m[8] - m[5] * m[7];
  const double k10 = m[2] * m[7] - m[1] * m[8];
  const double k20 = m[1] * m[5] - m[2] * m[4];
  const double det = k00 * m[0] + k10 * m[3] + k20 * m[6];
  const double id = 1.0 / det;
  /* result(i,j) = cofactor(j,i) * invdet */
  r[0] = k00 * id;
  r[1] = k10 * id;
  r[2] = k20 * id;
  r[3] = (m[5] * m[6] - m[3] * m[8]) * id;
  r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r[6] = (m[3] * m[7] - m[4] * m[6]) * id;
  r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* selfadjointView<Upper>().llt().solve(Identity) (LPV:114-115, 147-148, 180-181):
 * Cholesky of the symmetric matrix whose upper triangle is stored in a (n x n, row-major),
 * then two triangular solves per unit vector.  Writes the full inverse over a. */
static void llt_inverse_upper(int n, double* a) {
  double L[12 * 12];
  double X[12 * 12];
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    d = sqrt(d);
    L[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[j * n + i]; /* upper triangle: A(i,j) == A(j,i) stored at (j,i) */
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  for (int c = 0; c < n; ++c) {
    double y[12];
    for (int i = 0; i < n; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
      y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * X[k * n + c];
      X[i * n + c] = s / L[i * n + i];
    }
  }
  memcpy(a, X, sizeof(double) * (size_t)n * (size_t)n);
}

/* Least squares min ||G x - z|| for G (rows x 3), restating G.bdcSvd(ThinU|ThinV).solve(z)
 * (HLP:94) with Householder QR (identical for full-column-rank G up to rounding). */
static void lstsq3(int rows, double* G, double* z, double* x) {
  for (int j = 0; j < 3; ++j) {
    double nrm = 0;
    for (int i = j; i < rows; ++i) nrm += G[i * 3 + j] * G[i * 3 + j];
    nrm = sqrt(nrm);
    if (nrm == 0) continue;
    const double alpha = G[j * 3 + j] > 0 ? -nrm : nrm;
    const double v0 = G[j * 3 + j] - alpha;
    double vtv = v0 * v0;
    for (int i = j + 1; i < rows; ++i) vtv += G[i * 3 + j] * G[i * 3 + j];
    if (vtv > 0) {
      for (int c = j + 1; c < 3; ++c) {
        double s = v0 * G[j * 3 + c];
        for (int i = j + 1; i < rows; ++i) s += G[i * 3 + j] * G[i * 3 + c];
        s = 2 * s / vtv;
        G[j * 3 + c] -= s * v0;
        for (int i = j + 1; i < rows; ++i) G[i * 3 + c] -= s * G[i * 3 + j];
      }
      double s = v0 * z[j];
      for (int i = j + 1; i < rows; ++i) s += G[i * 3 + j] * z[i];
      s = 2 * s / vtv;
      z[j] -= s * v0;
      for (int i = j + 1; i < rows; ++i) z[i] -= s * G[i * 3 + j];
    }
    G[j * 3 + j] = alpha;
  }
  for (int i = 2; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 3; ++k) s -= G[i * 3 + k] * x[k];
    x[i] = s / G[i * 3 + i];
  }
}

/* ------------------------------------------------------------------------- */
/* L0: per-observation math                                                   */
/* ------------------------------------------------------------------------- */

/* compute_error_weight  HLP:52-74 */
void orc_error_weight(const orc_options* o, double res_squared, double* error, double* weight) {
  switch (o->robust_norm) {
    case ORC_NORM_HUBER: {
      const double thresh = o->huber_parameter;
      const double w = res_squared < thresh * thresh ? 1.0 : thresh / sqrt(res_squared);
      *error = 0.5 * (2 - w) * w * res_squared;
      *weight = w;
      return;
    }
    case ORC_NORM_CAUCHY:
      *error = log(1.0 + res_squared);
      *weight = 1.0;
      return;
    default:
      *error = 0.5 * res_squared;
      *weight = 1.0;
      return;
  }
}

/* linearize_point_pOSE  HLP:244-313 (and update_landmark_jacobian_pOSE HLP:383-454, same math).
 * res[4]; Jp[4x12] row-major or NULL; Jl[4x3] row-major or NULL. */
void orc_linearize_point_pose(double alpha, const double* obs, const double* x, const double* P,
                              double* res, double* Jp, double* Jl) {
  const double sa = sqrt(alpha);
  const double sb = sqrt(1.0 - alpha);
  double M[4][4];
  for (int j = 0; j < 4; ++j) {
    M[0][j] = sb * (P[j] - P[8 + j] * obs[0]);     /* HLP:251 */
    M[1][j] = sb * (P[4 + j] - P[8 + j] * obs[1]); /* HLP:252 */
    M[2][j] = sa * P[j];                           /* HLP:253 */
    M[3][j] = sa * P[4 + j];                       /* HLP:254 */
  }
  const double h[4] = {x[0], x[1], x[2], 1.0};
  for (int r = 0; r < 4; ++r) {
    double s = 0;
    for (int j = 0; j < 4; ++j) s += M[r][j] * h[j]; /* HLP:257 */
    res[r] = s;
  }
  res[2] -= sa * obs[0]; /* HLP:260 */
  res[3] -= sa * obs[1]; /* HLP:261 */
  if (Jp) {
    memset(Jp, 0, sizeof(double) * 48);
    for (int j = 0; j < 4; ++j) {
      Jp[0 * 12 + j] = h[j];                 /* HLP:273-276 */
      Jp[0 * 12 + 8 + j] = -h[j] * obs[0];   /* HLP:277-280 */
      Jp[1 * 12 + 4 + j] = h[j];             /* HLP:283-286 */
      Jp[1 * 12 + 8 + j] = -h[j] * obs[1];   /* HLP:287-290 */
      Jp[2 * 12 + j] = h[j];                 /* HLP:293-296 */
      Jp[3 * 12 + 4 + j] = h[j];             /* HLP:299-302 */
    }
    for (int j = 0; j < 12; ++j) {
      Jp[0 * 12 + j] *= sb; /* HLP:281 */
      Jp[1 * 12 + j] *= sb; /* HLP:291 */
      Jp[2 * 12 + j] *= sa; /* HLP:297 */
      Jp[3 * 12 + j] *= sa; /* HLP:303 */
    }
  }
  if (Jl) {
    for (int r = 0; r < 4; ++r)
      for (int j = 0; j < 3; ++j) Jl[r * 3 + j] = M[r][j]; /* HLP:310 */
  }
}

/* linearize_point_projective_space_homogeneous  HLP:316-380 with
 * BalCamera::project_projective_refinement_matrix_space_without_distortion
 * basalt_custom/camera/bal_camera.hpp:120-167.  res[2]; Jp[2x12]; Jl[2x4]. Returns validity. */
int orc_linearize_point_homogeneous(const double* obs, const double* X, const double* P,
                                    double* res, double* Jp, double* Jl) {
  double pc[3];
  for (int r = 0; r < 3; ++r) {
    double s = 0;
    for (int j = 0; j < 4; ++j) s += P[r * 4 + j] * X[j]; /* HLP:333 */
    pc[r] = s;
  }
  const double x = pc[0], y = pc[1], z = pc[2];
  res[0] = x / z - obs[0]; /* bal_camera.hpp:136-145, HLP:344 */
  res[1] = y / z - obs[1];
  const int valid = fabs(z) >= 1e-5; /* bal_camera.hpp:147, Sophus epsilonSqrt<double> */
  if (Jp || Jl) {
    const double d00 = 1 / z, d02 = -x / (z * z), d12 = -y / (z * z); /* bal_camera.hpp:153-160 */
    if (Jp) {
      for (int j = 0; j < 4; ++j) { /* HLP:352-369 */
        Jp[0 * 12 + j] = d00 * X[j];
        Jp[0 * 12 + 4 + j] = 0;
        Jp[0 * 12 + 8 + j] = d02 * X[j];
        Jp[1 * 12 + j] = 0;
        Jp[1 * 12 + 4 + j] = d00 * X[j];
        Jp[1 * 12 + 8 + j] = d12 * X[j];
      }
    }
    if (Jl) {
      for (int j = 0; j < 4; ++j) { /* HLP:376 */
        Jl[0 * 4 + j] = d00 * P[j] + d02 * P[8 + j];
        Jl[1 * 4 + j] = d00 * P[4 + j] + d12 * P[8 + j];
      }
    }
  }
  return valid;
}

static int all_finite(const double* v, int n) {
  for (int i = 0; i < n; ++i)
    if (!isfinite(v[i])) return 0;
  return 1;
}

/* ------------------------------------------------------------------------- */
/* step 1                                                                     */
/* ------------------------------------------------------------------------- */

/* initialize_varproj_lm_pOSE HLP:76-99 + initialize_varproj_pOSE HLP:221-241 (K1) */
void orc_init_landmarks_pose(const orc_problem* p, double alpha, const double* cams, double* lms) {
  const double sa = sqrt(alpha), sb = sqrt(1.0 - alpha);
  for (int l = 0; l < p->n_lms; ++l) {
    const int k = p->lm_off[l + 1] - p->lm_off[l];
    double* G = (double*)malloc(sizeof(double) * 12 * (size_t)k);
    double* z = (double*)malloc(sizeof(double) * 4 * (size_t)k);
    for (int i = 0; i < k; ++i) {
      const int o = p->lm_off[l] + i;
      const double* P = cams + 12 * (size_t)p->cam_idx[o];
      const double u = p->obs[2 * o], v = p->obs[2 * o + 1];
      for (int j = 0; j < 3; ++j) {
        G[(4 * i + 0) * 3 + j] = sb * (P[j] - P[8 + j] * u);     /* HLP:224 */
        G[(4 * i + 1) * 3 + j] = sb * (P[4 + j] - P[8 + j] * v); /* HLP:225 */
        G[(4 * i + 2) * 3 + j] = sa * P[j];                      /* HLP:226 */
        G[(4 * i + 3) * 3 + j] = sa * P[4 + j];                  /* HLP:227 */
      }
      z[4 * i + 0] = sb * (P[11] * u - P[3]); /* HLP:234 */
      z[4 * i + 1] = sb * (P[11] * v - P[7]); /* HLP:235 */
      z[4 * i + 2] = sa * (u - P[3]);         /* HLP:236 */
      z[4 * i + 3] = sa * (v - P[7]);         /* HLP:237 */
    }
    lstsq3(4 * k, G, z, lms + 3 * (size_t)l); /* HLP:94 */
    free(G);
    free(z);
  }
}

/* compute_error_pOSE HLP:117-154 + ResidualInfoAccu::add residual_info.cpp:96-110 (K2) */
void orc_error_pose(const orc_problem* p, const orc_options* o, double alpha, const double* cams,
                    const double* lms, orc_residual_info* out) {
  memset(out, 0, sizeof(*out));
  out->is_numerically_valid = 1;
  for (int l = 0; l < p->n_lms; ++l) {
    for (int i = p->lm_off[l]; i < p->lm_off[l + 1]; ++i) {
      double res[4], e, w;
      orc_linearize_point_pose(alpha, p->obs + 2 * (size_t)i, lms + 3 * (size_t)l,
                               cams + 12 * (size_t)p->cam_idx[i], res, NULL, NULL);
      const int nv = all_finite(res, 4);
      const double r2 = res[0] * res[0] + res[1] * res[1] + res[2] * res[2] + res[3] * res[3];
      orc_error_weight(o, r2, &e, &w);
      out->is_numerically_valid &= nv;
      out->all_num_obs += 1;
      out->all_error += e;
      out->all_residual_sum += sqrt(r2);
      /* projection_valid is always true on pOSE (HLP:263) */
      out->valid_num_obs += 1;
      out->valid_error += e;
      out->valid_residual_sum += sqrt(r2);
    }
  }
}

/* linearize_problem_pOSE LVP:63-82 -> linearize_landmark_pOSE LMB:135-178 (K3) */
int orc_linearize_pose(const orc_problem* p, const orc_options* o, double alpha, const double* cams,
                       const double* lms, double* storage) {
  int ok = 1;
  memset(storage, 0, sizeof(double) * 4 * ST * (size_t)p->n_obs); /* LMB:139 */
  for (int l = 0; l < p->n_lms; ++l) {
    for (int i = p->lm_off[l]; i < p->lm_off[l + 1]; ++i) {
      double res[4], Jp[48], Jl[12], e, w;
      orc_linearize_point_pose(alpha, p->obs + 2 * (size_t)i, lms + 3 * (size_t)l,
                               cams + 12 * (size_t)p->cam_idx[i], res, Jp, Jl);
      ok &= all_finite(res, 4) && all_finite(Jp, 48) && all_finite(Jl, 12); /* LMB:158-160 */
      const double r2 = res[0] * res[0] + res[1] * res[1] + res[2] * res[2] + res[3] * res[3];
      orc_error_weight(o, r2, &e, &w); /* LMB:162-165 */
      const double sw = sqrt(w);       /* LMB:166 */
      for (int r = 0; r < 4; ++r) {
        double* row = storage + ((size_t)4 * i + r) * ST;
        for (int j = 0; j < 12; ++j) row[j] = sw * Jp[r * 12 + j]; /* LMB:167 */
        for (int j = 0; j < 3; ++j) row[12 + j] = sw * Jl[r * 3 + j]; /* LMB:168 */
        row[15] = sw * res[r]; /* LMB:169 */
      }
    }
  }
  return ok;
}

/* get_Jp_diag2_pOSE LVP:183-222 -> add_Jp_diag2_pOSE LMB:272-282 (K4) */
void orc_jp_diag2_pose(const orc_problem* p, const double* storage, double* diag2) {
  memset(diag2, 0, sizeof(double) * 12 * (size_t)p->n_cams);
  for (int64_t i = 0; i < p->n_obs; ++i) {
    double* d = diag2 + 12 * (size_t)p->cam_idx[i];
    for (int j = 0; j < 12; ++j) {
      double s = 0;
      for (int r = 0; r < 4; ++r) {
        const double a = storage[((size_t)4 * i + r) * ST + j];
        s += a * a;
      }
      d[j] += s;
    }
  }
}

/* scale_Jl_cols_pOSE LVP:267-276 -> LMB:284-295 (K5) */
void orc_scale_jl_cols_pose(const orc_problem* p, const orc_options* o, double* storage,
                            double* jl_col_scale) {
  for (int l = 0; l < p->n_lms; ++l) {
    double n2[3] = {0, 0, 0};
    for (int64_t r = (int64_t)4 * p->lm_off[l]; r < (int64_t)4 * p->lm_off[l + 1]; ++r)
      for (int j = 0; j < 3; ++j) n2[j] += storage[r * ST + 12 + j] * storage[r * ST + 12 + j];
    double s[3];
    for (int j = 0; j < 3; ++j) {
      s[j] = 1.0 / (o->jacobi_scaling_eps + sqrt(n2[j])); /* LMB:289-292 */
      jl_col_scale[3 * (size_t)l + j] = s[j];
    }
    for (int64_t r = (int64_t)4 * p->lm_off[l]; r < (int64_t)4 * p->lm_off[l + 1]; ++r)
      for (int j = 0; j < 3; ++j) storage[r * ST + 12 + j] *= s[j]; /* LMB:294 */
  }
}

/* scale_Jp_cols_pOSE LVP:301-310 -> LMB:324-334 (K6) */
void orc_scale_jp_cols_pose(const orc_problem* p, double* storage, const double* scaling) {
  for (int64_t i = 0; i < p->n_obs; ++i) {
    const double* s = scaling + 12 * (size_t)p->cam_idx[i];
    for (int r = 0; r < 4; ++r)
      for (int j = 0; j < 12; ++j) storage[((size_t)4 * i + r) * ST + j] *= s[j];
  }
}

/* prepare_Hb_pOSE LPV:124-155 / prepare_Hb_pOSE_poBA LPV:157-188
 *  -> get_Hll_inv_add_Hpp_b_pOSE LMB:510-539 / _poBA LMB:541-572 (K7), then the per-camera
 *  damping + LLT inverse LPV:141-154 (K8).  lambda_lm = 0 reproduces the VarPro variant
 *  (no landmark damping); lambda_lm = lambda reproduces POWER_SCHUR_COMPLEMENT (LMB:549-550). */
void orc_prepare_hb_pose(const orc_problem* p, const double* storage, double lambda_pose,
                         double lambda_lm, double* hll_inv, double* b, double* b_inv) {
  memset(b, 0, sizeof(double) * 12 * (size_t)p->n_cams);       /* LPV:126 */
  memset(b_inv, 0, sizeof(double) * 144 * (size_t)p->n_cams);  /* LPV:125 */
  for (int l = 0; l < p->n_lms; ++l) {
    double H[9] = {0}, g[3] = {0};
    for (int64_t r = (int64_t)4 * p->lm_off[l]; r < (int64_t)4 * p->lm_off[l + 1]; ++r) {
      const double* row = storage + r * ST;
      for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) H[a * 3 + c] += row[12 + a] * row[12 + c]; /* LMB:517 */
        g[a] += row[12 + a] * row[15];                                          /* LMB:520 */
      }
    }
    H[0] += lambda_lm; H[4] += lambda_lm; H[8] += lambda_lm; /* LMB:550 (poBA only) */
    double* Hi = hll_inv + 9 * (size_t)l;
    inv3(H, Hi); /* LMB:518 */
    double w[3];
    for (int a = 0; a < 3; ++a) w[a] = Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2];
    for (int i = p->lm_off[l]; i < p->lm_off[l + 1]; ++i) {
      const int c = p->cam_idx[i];
      double e[4];
      for (int r = 0; r < 4; ++r) {
        const double* row = storage + ((size_t)4 * i + r) * ST;
        e[r] = row[15] - (row[12] * w[0] + row[13] * w[1] + row[14] * w[2]); /* LMB:529 */
      }
      for (int j = 0; j < 12; ++j) {
        double s = 0;
        for (int r = 0; r < 4; ++r) s += storage[((size_t)4 * i + r) * ST + j] * e[r];
        b[12 * (size_t)c + j] += s; /* LMB:533-534 */
        for (int k = 0; k < 12; ++k) {
          double h = 0;
          for (int r = 0; r < 4; ++r)
            h += storage[((size_t)4 * i + r) * ST + j] * storage[((size_t)4 * i + r) * ST + k];
          b_inv[144 * (size_t)c + 12 * j + k] += h; /* LMB:530, 535-536 */
        }
      }
    }
  }
  for (int c = 0; c < p->n_cams; ++c) {
    double* B = b_inv + 144 * (size_t)c;
    for (int j = 0; j < 12; ++j) B[13 * j] += lambda_pose; /* LPV:146 */
    llt_inverse_upper(12, B);                              /* LPV:147-148 */
  }
}

/* right_mul_b_inv_pOSE LPV:322-340 / right_mul_b_inv_joint LPV:342-360 (K9) */
void orc_right_mul_b_inv(int32_t n_cams, int32_t dim, const double* b_inv, const double* x, double* y) {
  for (int c = 0; c < n_cams; ++c)
    for (int r = 0; r < dim; ++r) {
      double s = 0;
      for (int j = 0; j < dim; ++j)
        s += b_inv[(size_t)dim * dim * c + dim * r + j] * x[(size_t)dim * c + j];
      y[(size_t)dim * c + r] = s;
    }
}

/* body of right_mul_e0_pOSE for one landmark, LPV:369-398.  lock == NULL: no mutex. */
static void e0_landmark_pose(const orc_problem* p, const double* storage, const double* hll_inv,
                             const double* x, double* y, int l, pthread_mutex_t* locks) {
  const int b = p->lm_off[l], e = p->lm_off[l + 1], k = e - b;
  double stack_buf[4 * 64];
  double* jp_x = k <= 64 ? stack_buf : (double*)malloc(sizeof(double) * 4 * (size_t)k); /* LPV:374 */
  double u[3] = {0, 0, 0};
  for (int i = b; i < e; ++i) {
    const double* xc = x + 12 * (size_t)p->cam_idx[i];
    for (int r = 0; r < 4; ++r) {
      const double* row = storage + ((size_t)4 * i + r) * ST;
      double s = 0;
      for (int j = 0; j < 12; ++j) s += row[j] * xc[j]; /* LPV:377-380 */
      jp_x[4 * (i - b) + r] = s;
    }
  }
  for (int i = b; i < e; ++i)
    for (int r = 0; r < 4; ++r) {
      const double* row = storage + ((size_t)4 * i + r) * ST;
      for (int a = 0; a < 3; ++a) u[a] += row[12 + a] * jp_x[4 * (i - b) + r]; /* jl^T jp_x LPV:385 */
    }
  const double* Hi = hll_inv + 9 * (size_t)l;
  double v[3];
  for (int a = 0; a < 3; ++a) v[a] = Hi[a * 3] * u[0] + Hi[a * 3 + 1] * u[1] + Hi[a * 3 + 2] * u[2];
  for (int i = b; i < e; ++i) {
    const int c = p->cam_idx[i];
    double s[4], out[12];
    for (int r = 0; r < 4; ++r) {
      const double* row = storage + ((size_t)4 * i + r) * ST;
      s[r] = row[12] * v[0] + row[13] * v[1] + row[14] * v[2]; /* jl * (...) LPV:385 */
    }
    for (int j = 0; j < 12; ++j) {
      double t = 0;
      for (int r = 0; r < 4; ++r) t += storage[((size_t)4 * i + r) * ST + j] * s[r]; /* LPV:396 */
      out[j] = t;
    }
    if (locks) pthread_mutex_lock(&locks[c]); /* LPV:394 */
    for (int j = 0; j < 12; ++j) y[12 * (size_t)c + j] += out[j]; /* LPV:395 */
    if (locks) pthread_mutex_unlock(&locks[c]);
  }
  if (jp_x != stack_buf) free(jp_x);
}

/* right_mul_e0_pOSE LPV:364-406 (K10), single thread */
void orc_right_mul_e0_pose(const orc_problem* p, const double* storage, const double* hll_inv,
                           const double* x, double* y) {
  memset(y, 0, sizeof(double) * 12 * (size_t)p->n_cams); /* LPV:367 */
  for (int l = 0; l < p->n_lms; ++l) e0_landmark_pose(p, storage, hll_inv, x, y, l, NULL);
}

typedef struct {
  const orc_problem* p;
  const double* storage;
  const double* hll_inv;
  const double* x;
  double* y;
  int l0, l1;
  pthread_mutex_t* locks;
  int* next;  /* dynamic schedule: shared chunk cursor (NULL = this job's static range [l0, l1)) */
  int grain;
} e0_job;

static void* e0_worker(void* arg) {
  e0_job* j = (e0_job*)arg;
  if (j->next) {
    /* tbb::parallel_for's default auto_partitioner hands out sub-ranges on demand (work stealing): restated as
     * chunks of `grain` landmarks taken from one shared cursor */
    for (;;) {
      const int l0 = __atomic_fetch_add(j->next, j->grain, __ATOMIC_RELAXED);
      if (l0 >= j->p->n_lms) break;
      const int l1 = l0 + j->grain < j->p->n_lms ? l0 + j->grain : j->p->n_lms;
      for (int l = l0; l < l1; ++l) e0_landmark_pose(j->p, j->storage, j->hll_inv, j->x, j->y, l, j->locks);
    }
    return NULL;
  }
  for (int l = j->l0; l < j->l1; ++l)
    e0_landmark_pose(j->p, j->storage, j->hll_inv, j->x, j->y, l, j->locks);
  return NULL;
}

/* 0: contiguous landmark ranges balanced by observation count, one per thread (tbb static_partitioner analogue);
 * g > 0: chunks of g landmarks on demand (auto_partitioner analogue, LPV:402-403 uses TBB's default). */
static int g_e0_grain = 0;
void orc_set_e0_schedule(int32_t grain) { g_e0_grain = grain > 0 ? grain : 0; }
/* How the threads combine their `res += Jp^T s`.  0: the reference's scheme -- one std::mutex per camera, taken per
 * observation (LPV:393-397).  1: every thread adds into a result vector of its own and the vectors are summed after the
 * join -- what the reference's Reductor does for the per-camera column norms (linearization_varproj.hpp:184-209), NOT
 * what it does here; a second baseline that shows how much of the multithreaded slowdown on a hub-heavy graph is the
 * mutex and how much is this restatement (bench.py cpu_baseline.threads_tried). */
static int g_e0_private = 0;
void orc_set_e0_scatter(int32_t private_sums) { g_e0_private = private_sums != 0; }

/* persistent worker pool (the reference runs on TBB's pool: thread creation is not part of its
 * per-term cost).  Workers park on a barrier pair; one pool per thread count, never torn down. */
typedef struct {
  int n;
  pthread_t* th;
  e0_job* jobs;
  pthread_barrier_t start, done;
} e0_pool;

typedef struct {
  e0_pool* pool;
  int idx;
} e0_pool_arg;

static void* e0_pool_worker(void* arg) {
  e0_pool_arg* a = (e0_pool_arg*)arg;
  for (;;) {
    pthread_barrier_wait(&a->pool->start);
    e0_worker(&a->pool->jobs[a->idx]);
    pthread_barrier_wait(&a->pool->done);
  }
  return NULL;
}

static e0_pool* get_pool(int n_threads) {
  static e0_pool* pools[8];
  static int n_pools = 0;
  for (int i = 0; i < n_pools; ++i)
    if (pools[i]->n == n_threads) return pools[i];
  if (n_pools == 8) return NULL;
  e0_pool* p = (e0_pool*)calloc(1, sizeof(e0_pool));
  p->n = n_threads;
  p->th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
  p->jobs = (e0_job*)calloc((size_t)n_threads, sizeof(e0_job));
  pthread_barrier_init(&p->start, NULL, (unsigned)n_threads + 1);
  pthread_barrier_init(&p->done, NULL, (unsigned)n_threads + 1);
  for (int t = 0; t < n_threads; ++t) {
    e0_pool_arg* a = (e0_pool_arg*)malloc(sizeof(e0_pool_arg));
    a->pool = p;
    a->idx = t;
    pthread_create(&p->th[t], NULL, e0_pool_worker, a);
  }
  pools[n_pools++] = p;
  return p;
}

/* right_mul_e0_pOSE with the reference's parallel structure: tbb::parallel_for over
 * landmark ranges (LPV:402-403) restated as contiguous ranges balanced by observation
 * count on n_threads pooled pthreads; std::scoped_lock(pose_mutex_[c]) per observation
 * (LPV:393-397). */
void orc_right_mul_e0_pose_mt(const orc_problem* p, const double* storage, const double* hll_inv,
                              const double* x, double* y, int32_t n_threads) {
  e0_pool* pool = n_threads > 1 ? get_pool(n_threads) : NULL;
  if (!pool) {
    orc_right_mul_e0_pose(p, storage, hll_inv, x, y);
    return;
  }
  memset(y, 0, sizeof(double) * 12 * (size_t)p->n_cams);
  static pthread_mutex_t* locks = NULL;
  static int n_locks = 0;
  if (n_locks < p->n_cams) {
    locks = (pthread_mutex_t*)realloc(locks, sizeof(pthread_mutex_t) * (size_t)p->n_cams);
    for (int c = n_locks; c < p->n_cams; ++c) pthread_mutex_init(&locks[c], NULL);
    n_locks = p->n_cams;
  }
  int l = 0;
  static int cursor;
  cursor = 0;
  const size_t n = 12 * (size_t)p->n_cams;
  static double* priv = NULL;  /* [n_threads][n] private sums (scatter scheme 1) */
  static size_t priv_size = 0;
  if (g_e0_private && priv_size < n * (size_t)n_threads) {
    priv = (double*)realloc(priv, sizeof(double) * n * (size_t)n_threads);
    priv_size = n * (size_t)n_threads;
  }
  if (g_e0_private) memset(priv, 0, sizeof(double) * n * (size_t)n_threads);
  for (int t = 0; t < n_threads; ++t) {
    const int64_t target = p->n_obs * (int64_t)(t + 1) / n_threads;
    int l1 = l;
    while (l1 < p->n_lms && p->lm_off[l1 + 1] <= target) ++l1;
    if (t == n_threads - 1) l1 = p->n_lms;
    pool->jobs[t] = (e0_job){p, storage, hll_inv, x, g_e0_private ? priv + n * (size_t)t : y, l, l1,
                             g_e0_private ? NULL : locks, g_e0_grain > 0 ? &cursor : NULL, g_e0_grain};
    l = l1;
  }
  pthread_barrier_wait(&pool->start);
  pthread_barrier_wait(&pool->done);
  if (g_e0_private)
    for (int t = 0; t < n_threads; ++t)
      for (size_t i = 0; i < n; ++i) y[i] += priv[n * (size_t)t + i];
}

static double norm2(const double* v, size_t n) {
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}

/* solve_pOSE LPV:191-237 (K9+K10+K11).  terms (optional) receives the m+1 series terms
 * [ (m+1) x 12 n_cams ] for term-by-term parity tests.  Returns the termination type. */
int orc_solve_pose(const orc_problem* p, const double* storage, const double* hll_inv,
                   const double* b_inv, const double* b, int32_t m, double q_tol, double r_tol,
                   double* accum, int32_t* num_iterations, double* terms, int32_t n_threads) {
  const size_t n = 12 * (size_t)p->n_cams;
  double* tmp = (double*)malloc(sizeof(double) * n);
  double* y = (double*)malloc(sizeof(double) * n);
  for (size_t i = 0; i < n; ++i) y[i] = -b[i];
  orc_right_mul_b_inv(p->n_cams, 12, b_inv, y, accum); /* LPV:196 */
  if (terms) memcpy(terms, accum, sizeof(double) * n);
  int status = ORC_NO_CONVERGENCE;
  *num_iterations = m; /* LPV:233-236 */
  if (m > 0) {
    const double norm_0 = r_tol > 0 ? norm2(accum, n) : 0; /* LPV:198 */
    memcpy(tmp, accum, sizeof(double) * n);                /* LPV:200 */
    for (int i = 1; i <= m; ++i) {
      orc_right_mul_e0_pose_mt(p, storage, hll_inv, tmp, y, n_threads);
      orc_right_mul_b_inv(p->n_cams, 12, b_inv, y, tmp); /* LPV:202 */
      for (size_t k = 0; k < n; ++k) accum[k] += tmp[k]; /* LPV:203 */
      if (terms) memcpy(terms + (size_t)i * n, tmp, sizeof(double) * n);
      const double iter_norm = (q_tol > 0 || r_tol > 0) ? norm2(tmp, n) : 0; /* LPV:206-207 */
      if (q_tol > 0) {
        const double zeta = i * iter_norm / norm2(accum, n); /* LPV:209 */
        if (zeta < q_tol) {
          status = ORC_SUCCESS;
          *num_iterations = i;
          break;
        }
      }
      if (r_tol > 0 && iter_norm / norm_0 < r_tol) { /* LPV:220 */
        status = ORC_SUCCESS;
        *num_iterations = i;
        break;
      }
    }
  }
  free(tmp);
  free(y);
  return status;
}

/* back_substitute_pOSE LPV:289-302 -> LMB:670-707 (K12, POWER_VARPROJ).  cams = cameras AFTER
 * the pose update (LZR:251-256); inc = pose increment after the scale/unscale round trip. */
double orc_back_substitute_pose(const orc_problem* p, double alpha, const double* storage,
                                const double* cams, double* lms, const double* inc) {
  double l_diff = 0;
  for (int l = 0; l < p->n_lms; ++l) {
    const int b = p->lm_off[l], e = p->lm_off[l + 1], k = e - b;
    double H[9] = {0}, g[3] = {0};
    double* J_inc = (double*)calloc(4 * (size_t)k, sizeof(double)); /* LMB:675-676 */
    for (int i = b; i < e; ++i) {
      const int c = p->cam_idx[i];
      double res[4], Jp[48], Jl[12];
      orc_linearize_point_pose(alpha, p->obs + 2 * (size_t)i, lms + 3 * (size_t)l,
                               cams + 12 * (size_t)c, res, Jp, Jl); /* LMB:686-687 */
      for (int r = 0; r < 4; ++r) {
        for (int a = 0; a < 3; ++a) {
          for (int d = 0; d < 3; ++d) H[a * 3 + d] += Jl[r * 3 + a] * Jl[r * 3 + d]; /* LMB:694 */
          g[a] += Jl[r * 3 + a] * res[r];                                            /* LMB:697 */
        }
        double s = 0;
        for (int j = 0; j < 12; ++j) s += Jp[r * 12 + j] * inc[12 * (size_t)c + j]; /* LMB:699 */
        J_inc[4 * (i - b) + r] += s;
      }
    }
    double Hi[9], d[3];
    inv3(H, Hi);
    for (int a = 0; a < 3; ++a)
      d[a] = -(Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2]); /* LMB:702 */
    for (int i = b; i < e; ++i)
      for (int r = 0; r < 4; ++r) {
        const double* row = storage + ((size_t)4 * i + r) * ST;
        J_inc[4 * (i - b) + r] += row[12] * d[0] + row[13] * d[1] + row[14] * d[2]; /* LMB:704 */
      }
    double s = 0;
    for (int i = b; i < e; ++i)
      for (int r = 0; r < 4; ++r) {
        const double ji = J_inc[4 * (i - b) + r];
        s += ji * (0.5 * ji + storage[((size_t)4 * i + r) * ST + 15]); /* LMB:705 */
      }
    l_diff -= s;
    for (int a = 0; a < 3; ++a) lms[3 * (size_t)l + a] += d[a]; /* LMB:706 */
    free(J_inc);
  }
  return l_diff;
}

/* back_substitute_poBA LVP:168-181 -> LMB:625-656 (K12, POWER_SCHUR_COMPLEMENT) */
double orc_back_substitute_poba(const orc_problem* p, const double* storage,
                                const double* jl_col_scale, double lambda_lm, double* lms,
                                const double* inc) {
  double l_diff = 0;
  for (int l = 0; l < p->n_lms; ++l) {
    const int b = p->lm_off[l], e = p->lm_off[l + 1], k = e - b;
    double H[9] = {0}, g[3] = {0};
    double* J_inc = (double*)calloc(4 * (size_t)k, sizeof(double));
    for (int i = b; i < e; ++i) {
      const int c = p->cam_idx[i];
      for (int r = 0; r < 4; ++r) {
        const double* row = storage + ((size_t)4 * i + r) * ST;
        double s = 0;
        for (int j = 0; j < 12; ++j) s += row[j] * inc[12 * (size_t)c + j]; /* LMB:643 */
        J_inc[4 * (i - b) + r] += s;
        for (int a = 0; a < 3; ++a) {
          for (int d = 0; d < 3; ++d) H[a * 3 + d] += row[12 + a] * row[12 + d]; /* LMB:640 */
          g[a] += row[12 + a] * (row[15] + s);                                    /* LMB:642 */
        }
      }
    }
    H[0] += lambda_lm; H[4] += lambda_lm; H[8] += lambda_lm; /* LMB:646 */
    double Hi[9], d[3];
    inv3(H, Hi);
    for (int a = 0; a < 3; ++a)
      d[a] = -(Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2]); /* LMB:647 */
    double s = 0;
    for (int i = b; i < e; ++i)
      for (int r = 0; r < 4; ++r) {
        const double* row = storage + ((size_t)4 * i + r) * ST;
        const double ji = J_inc[4 * (i - b) + r] + row[12] * d[0] + row[13] * d[1] + row[14] * d[2];
        s += ji * (0.5 * ji + row[15]); /* LMB:649-651 */
      }
    l_diff -= s;
    for (int a = 0; a < 3; ++a)
      lms[3 * (size_t)l + a] += d[a] * jl_col_scale[3 * (size_t)l + a]; /* LMB:653-654 */
    free(J_inc);
  }
  return l_diff;
}

/* Camera::apply_inc_pose_pOSE bal_problem.hpp:147-157 (K13); inc already multiplied by the
 * pose Jacobian scaling (LZR:251-254). */
void orc_apply_cam_inc(int32_t n_cams, double* cams, const double* inc) {
  for (size_t i = 0; i < 12 * (size_t)n_cams; ++i) cams[i] += inc[i];
}

/* ------------------------------------------------------------------------- */
/* step 2                                                                     */
/* ------------------------------------------------------------------------- */

/* kernel_COD HLP:202-216: an orthonormal basis N (n x (n-1), row-major) of null(v^T).
 * Eigen's CompleteOrthogonalDecomposition output is not reproducible without Eigen; every
 * orthonormal basis gives the same ambient-space quantities, so one Householder
 * reflector is used: H = I - 2 w w^T / (w^T w), w = v + sign(v0)|v| e0, N = H[:, 1:]. */
void orc_kernel_basis(int32_t n, const double* v, double* N) {
  double w[16];
  double nv = 0;
  for (int i = 0; i < n; ++i) nv += v[i] * v[i];
  nv = sqrt(nv);
  for (int i = 0; i < n; ++i) w[i] = v[i];
  w[0] += v[0] >= 0 ? nv : -nv;
  double wtw = 0;
  for (int i = 0; i < n; ++i) wtw += w[i] * w[i];
  for (int i = 0; i < n; ++i)
    for (int j = 1; j < n; ++j)
      N[i * (n - 1) + (j - 1)] = (i == j ? 1.0 : 0.0) - 2.0 * w[i] * w[j] / wtw;
}

/* compute_error_projective_space_homogeneous HLP:157-196 (K2') */
void orc_error_homogeneous(const orc_problem* p, const orc_options* o, const double* cams,
                           const double* lms_h, orc_residual_info* out) {
  memset(out, 0, sizeof(*out));
  out->is_numerically_valid = 1;
  for (int l = 0; l < p->n_lms; ++l) {
    for (int i = p->lm_off[l]; i < p->lm_off[l + 1]; ++i) {
      double res[2], e, w;
      const int valid = orc_linearize_point_homogeneous(
          p->obs + 2 * (size_t)i, lms_h + 4 * (size_t)l, cams + 12 * (size_t)p->cam_idx[i], res,
          NULL, NULL);
      const int nv = all_finite(res, 2);
      const double r2 = res[0] * res[0] + res[1] * res[1];
      orc_error_weight(o, r2, &e, &w);
      out->is_numerically_valid &= nv;
      out->all_num_obs += 1;
      out->all_error += e;
      out->all_residual_sum += sqrt(r2);
      if (valid) {
        out->valid_num_obs += 1;
        out->valid_error += e;
        out->valid_residual_sum += sqrt(r2);
      }
    }
  }
}

/* linearize_problem_projective_space_homogeneous LVP:84-103 -> LMB:180-225 (K3') */
int orc_linearize_homogeneous(const orc_problem* p, const orc_options* o, const double* cams,
                              const double* lms_h, double* storage_h) {
  int ok = 1;
  memset(storage_h, 0, sizeof(double) * 2 * SH * (size_t)p->n_obs);
  for (int l = 0; l < p->n_lms; ++l) {
    for (int i = p->lm_off[l]; i < p->lm_off[l + 1]; ++i) {
      double res[2], Jp[24], Jl[8], e, w;
      orc_linearize_point_homogeneous(p->obs + 2 * (size_t)i, lms_h + 4 * (size_t)l,
                                      cams + 12 * (size_t)p->cam_idx[i], res, Jp, Jl);
      ok &= all_finite(res, 2) && all_finite(Jp, 24) && all_finite(Jl, 8);
      const double r2 = res[0] * res[0] + res[1] * res[1];
      orc_error_weight(o, r2, &e, &w);
      const double sw = sqrt(w);
      for (int r = 0; r < 2; ++r) {
        double* row = storage_h + ((size_t)2 * i + r) * SH;
        for (int j = 0; j < 12; ++j) row[j] = sw * Jp[r * 12 + j];    /* LMB:214 */
        for (int j = 0; j < 4; ++j) row[12 + j] = sw * Jl[r * 4 + j]; /* LMB:215 */
        row[16] = sw * res[r];                                        /* LMB:216 */
      }
    }
  }
  return ok;
}

/* get_Jp_diag2_projective_space LVP:225-264 -> LMB:658-668 (K4'); only the first 12 n_cams
 * entries of the reference's 16 n_cams vector are ever written or read (LVP:253 quirk). */
void orc_jp_diag2_homogeneous(const orc_problem* p, const double* storage_h, double* diag2) {
  memset(diag2, 0, sizeof(double) * 12 * (size_t)p->n_cams);
  for (int64_t i = 0; i < p->n_obs; ++i) {
    double* d = diag2 + 12 * (size_t)p->cam_idx[i];
    for (int j = 0; j < 12; ++j) {
      double s = 0;
      for (int r = 0; r < 2; ++r) {
        const double a = storage_h[((size_t)2 * i + r) * SH + j];
        s += a * a;
      }
      d[j] += s;
    }
  }
}

/* scale_Jl_cols_homogeneous LVP:278-287 -> LMB:298-309 (K5') */
void orc_scale_jl_cols_homogeneous(const orc_problem* p, const orc_options* o, double* storage_h,
                                   double* jl_col_scale_h) {
  for (int l = 0; l < p->n_lms; ++l) {
    double n2[4] = {0, 0, 0, 0};
    for (int64_t r = (int64_t)2 * p->lm_off[l]; r < (int64_t)2 * p->lm_off[l + 1]; ++r)
      for (int j = 0; j < 4; ++j) n2[j] += storage_h[r * SH + 12 + j] * storage_h[r * SH + 12 + j];
    double s[4];
    for (int j = 0; j < 4; ++j) {
      s[j] = 1.0 / (o->jacobi_scaling_eps + sqrt(n2[j]));
      jl_col_scale_h[4 * (size_t)l + j] = s[j];
    }
    for (int64_t r = (int64_t)2 * p->lm_off[l]; r < (int64_t)2 * p->lm_off[l + 1]; ++r)
      for (int j = 0; j < 4; ++j) storage_h[r * SH + 12 + j] *= s[j];
  }
}

/* scale_Jp_cols_joint LVP:290-299 -> LMB:311-321 (K6') */
void orc_scale_jp_cols_joint(const orc_problem* p, double* storage_h, const double* scaling) {
  for (int64_t i = 0; i < p->n_obs; ++i) {
    const double* s = scaling + 12 * (size_t)p->cam_idx[i];
    for (int r = 0; r < 2; ++r)
      for (int j = 0; j < 12; ++j) storage_h[((size_t)2 * i + r) * SH + j] *= s[j];
  }
}

/* linearize_nullspace LVP:105-124 -> LMB:227-269 (K6b) */
void orc_linearize_nullspace(const orc_problem* p, const double* cams, const double* lms_h,
                             const double* storage_h, double* storage_n) {
  double* Nc = (double*)malloc(sizeof(double) * 132 * (size_t)p->n_cams);
  for (int c = 0; c < p->n_cams; ++c) orc_kernel_basis(12, cams + 12 * (size_t)c, Nc + 132 * (size_t)c);
  for (int l = 0; l < p->n_lms; ++l) {
    double Nl[12];
    orc_kernel_basis(4, lms_h + 4 * (size_t)l, Nl); /* LMB:234 */
    for (int i = p->lm_off[l]; i < p->lm_off[l + 1]; ++i) {
      const double* N = Nc + 132 * (size_t)p->cam_idx[i]; /* LMB:257 */
      for (int r = 0; r < 2; ++r) {
        const double* row = storage_h + ((size_t)2 * i + r) * SH;
        double* out = storage_n + ((size_t)2 * i + r) * SN;
        for (int j = 0; j < 11; ++j) {
          double s = 0;
          for (int k = 0; k < 12; ++k) s += row[k] * N[k * 11 + j]; /* LMB:259 */
          out[j] = s;
        }
        for (int j = 0; j < 3; ++j) {
          double s = 0;
          for (int k = 0; k < 4; ++k) s += row[12 + k] * Nl[k * 3 + j]; /* LMB:260 */
          out[11 + j] = s;
        }
      }
    }
  }
  free(Nc);
}

/* prepare_Hb_joint LPV:74-122 -> get_Hll_inv_add_Hpp_b_joint LMB:474-507 (K7'), then the
 * per-camera damping Proj_pose^T lambda Proj_pose + LLT inverse LPV:91-121 (K8').
 * NOTE: the landmark damping Proj^T lambda Proj (LMB:485) uses the landmark's CURRENT
 * p_w_homogeneous, which only enters through Proj^T Proj = I_3; it is restated as lambda*I_3
 * evaluated the same way (N^T (lambda N)). */
void orc_prepare_hb_joint(const orc_problem* p, const double* storage_h, const double* storage_n,
                          double lambda, double* hll_inv, double* b, double* b_inv) {
  memset(b, 0, sizeof(double) * 11 * (size_t)p->n_cams);
  memset(b_inv, 0, sizeof(double) * 121 * (size_t)p->n_cams);
  for (int l = 0; l < p->n_lms; ++l) {
    double H[9] = {0}, g[3] = {0};
    for (int64_t r = (int64_t)2 * p->lm_off[l]; r < (int64_t)2 * p->lm_off[l + 1]; ++r) {
      const double* row = storage_n + r * SN;
      const double res = storage_h[r * SH + 16];
      for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c) H[a * 3 + c] += row[11 + a] * row[11 + c]; /* LMB:484 */
        g[a] += row[11 + a] * res;                                              /* LMB:487 */
      }
    }
    H[0] += lambda; H[4] += lambda; H[8] += lambda; /* LMB:485: Proj^T lambda Proj == lambda I_3 */
    double* Hi = hll_inv + 9 * (size_t)l;
    inv3(H, Hi); /* LMB:486 */
    double w[3];
    for (int a = 0; a < 3; ++a) w[a] = Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2];
    for (int i = p->lm_off[l]; i < p->lm_off[l + 1]; ++i) {
      const int c = p->cam_idx[i];
      double e[2];
      for (int r = 0; r < 2; ++r) {
        const double* row = storage_n + ((size_t)2 * i + r) * SN;
        e[r] = storage_h[((size_t)2 * i + r) * SH + 16] -
               (row[11] * w[0] + row[12] * w[1] + row[13] * w[2]); /* LMB:497 */
      }
      for (int j = 0; j < 11; ++j) {
        const double j0 = storage_n[((size_t)2 * i) * SN + j];
        const double j1 = storage_n[((size_t)2 * i + 1) * SN + j];
        b[11 * (size_t)c + j] += j0 * e[0] + j1 * e[1]; /* LMB:501-502 */
        for (int k = 0; k < 11; ++k)
          b_inv[121 * (size_t)c + 11 * j + k] += j0 * storage_n[((size_t)2 * i) * SN + k] +
                                                  j1 * storage_n[((size_t)2 * i + 1) * SN + k]; /* LMB:498, 503-504 */
      }
    }
  }
  for (int c = 0; c < p->n_cams; ++c) {
    double* B = b_inv + 121 * (size_t)c;
    for (int j = 0; j < 11; ++j) B[12 * j] += lambda; /* LPV:112-113: Proj_pose^T lambda Proj_pose == lambda I_11 */
    llt_inverse_upper(11, B);                         /* LPV:114-115 */
  }
}

/* right_mul_e0_joint LPV:408-453 (K10') */
void orc_right_mul_e0_joint(const orc_problem* p, const double* storage_n, const double* hll_inv,
                            const double* x, double* y) {
  memset(y, 0, sizeof(double) * 11 * (size_t)p->n_cams);
  for (int l = 0; l < p->n_lms; ++l) {
    const int b = p->lm_off[l], e = p->lm_off[l + 1];
    double u[3] = {0, 0, 0};
    for (int i = b; i < e; ++i) {
      const double* xc = x + 11 * (size_t)p->cam_idx[i];
      for (int r = 0; r < 2; ++r) {
        const double* row = storage_n + ((size_t)2 * i + r) * SN;
        double s = 0;
        for (int j = 0; j < 11; ++j) s += row[j] * xc[j]; /* LPV:424-428 */
        for (int a = 0; a < 3; ++a) u[a] += row[11 + a] * s; /* LPV:432 */
      }
    }
    const double* Hi = hll_inv + 9 * (size_t)l;
    double v[3];
    for (int a = 0; a < 3; ++a) v[a] = Hi[a * 3] * u[0] + Hi[a * 3 + 1] * u[1] + Hi[a * 3 + 2] * u[2];
    for (int i = b; i < e; ++i) {
      const int c = p->cam_idx[i];
      const double* r0 = storage_n + ((size_t)2 * i) * SN;
      const double* r1 = r0 + SN;
      const double s0 = r0[11] * v[0] + r0[12] * v[1] + r0[13] * v[2];
      const double s1 = r1[11] * v[0] + r1[12] * v[1] + r1[13] * v[2];
      for (int j = 0; j < 11; ++j) y[11 * (size_t)c + j] += r0[j] * s0 + r1[j] * s1; /* LPV:442-443 */
    }
  }
}

/* solve_joint LPV:240-287 */
int orc_solve_joint(const orc_problem* p, const double* storage_n, const double* hll_inv,
                    const double* b_inv, const double* b, int32_t m, double q_tol, double r_tol,
                    double* accum, int32_t* num_iterations, double* terms) {
  const size_t n = 11 * (size_t)p->n_cams;
  double* tmp = (double*)malloc(sizeof(double) * n);
  double* y = (double*)malloc(sizeof(double) * n);
  for (size_t i = 0; i < n; ++i) y[i] = -b[i];
  orc_right_mul_b_inv(p->n_cams, 11, b_inv, y, accum); /* LPV:246 */
  if (terms) memcpy(terms, accum, sizeof(double) * n);
  int status = ORC_NO_CONVERGENCE;
  *num_iterations = m;
  if (m > 0) {
    const double norm_0 = r_tol > 0 ? norm2(accum, n) : 0;
    memcpy(tmp, accum, sizeof(double) * n);
    for (int i = 1; i <= m; ++i) {
      orc_right_mul_e0_joint(p, storage_n, hll_inv, tmp, y);
      orc_right_mul_b_inv(p->n_cams, 11, b_inv, y, tmp); /* LPV:252 */
      for (size_t k = 0; k < n; ++k) accum[k] += tmp[k];
      if (terms) memcpy(terms + (size_t)i * n, tmp, sizeof(double) * n);
      const double iter_norm = (q_tol > 0 || r_tol > 0) ? norm2(tmp, n) : 0;
      if (q_tol > 0) {
        const double zeta = i * iter_norm / norm2(accum, n);
        if (zeta < q_tol) {
          status = ORC_SUCCESS;
          *num_iterations = i;
          break;
        }
      }
      if (r_tol > 0 && iter_norm / norm_0 < r_tol) {
        status = ORC_SUCCESS;
        *num_iterations = i;
        break;
      }
    }
  }
  free(tmp);
  free(y);
  return status;
}

/* back_substitute_joint LVP:151-165 -> LMB:574-623 (K12').  cams = cameras BEFORE the pose
 * update (LZR:280 runs before LZR:283-305). */
double orc_back_substitute_joint(const orc_problem* p, const double* storage_h,
                                 const double* jl_col_scale_h, double lambda, const double* cams,
                                 double* lms_h, const double* inc) {
  double l_diff = 0;
  double* Nc = (double*)malloc(sizeof(double) * 132 * (size_t)p->n_cams);
  double* pinc = (double*)malloc(sizeof(double) * 12 * (size_t)p->n_cams);
  for (int c = 0; c < p->n_cams; ++c) {
    double* N = Nc + 132 * (size_t)c;
    orc_kernel_basis(12, cams + 12 * (size_t)c, N); /* LMB:601 */
    for (int k = 0; k < 12; ++k) {
      double s = 0;
      for (int j = 0; j < 11; ++j) s += N[k * 11 + j] * inc[11 * (size_t)c + j]; /* Proj_pose * p_inc */
      pinc[12 * (size_t)c + k] = s;
    }
  }
  for (int l = 0; l < p->n_lms; ++l) {
    const int b = p->lm_off[l], e = p->lm_off[l + 1], k = e - b;
    double Nl[12];
    orc_kernel_basis(4, lms_h + 4 * (size_t)l, Nl); /* LMB:581 */
    double H[9] = {0}, g[3] = {0};
    double* J_inc = (double*)calloc(2 * (size_t)k, sizeof(double));
    for (int i = b; i < e; ++i) {
      const int c = p->cam_idx[i];
      for (int r = 0; r < 2; ++r) {
        const double* row = storage_h + ((size_t)2 * i + r) * SH;
        double jlp[3];
        for (int a = 0; a < 3; ++a) {
          double s = 0;
          for (int q = 0; q < 4; ++q) s += row[12 + q] * Nl[q * 3 + a]; /* LMB:606 */
          jlp[a] = s;
        }
        double jpi = 0;
        for (int j = 0; j < 12; ++j) jpi += row[j] * pinc[12 * (size_t)c + j]; /* LMB:609-610 */
        for (int a = 0; a < 3; ++a) {
          for (int d = 0; d < 3; ++d) H[a * 3 + d] += jlp[a] * jlp[d]; /* LMB:607 */
          g[a] += jlp[a] * (row[16] + jpi);                            /* LMB:609 */
        }
        J_inc[2 * (i - b) + r] += jpi; /* LMB:610 */
      }
    }
    H[0] += lambda; H[4] += lambda; H[8] += lambda; /* LMB:613 */
    double Hi[9], d[3], dp[4];
    inv3(H, Hi);
    for (int a = 0; a < 3; ++a)
      d[a] = -(Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2]); /* LMB:614 */
    for (int q = 0; q < 4; ++q) dp[q] = Nl[q * 3] * d[0] + Nl[q * 3 + 1] * d[1] + Nl[q * 3 + 2] * d[2]; /* LMB:615 */
    double s = 0;
    for (int i = b; i < e; ++i)
      for (int r = 0; r < 2; ++r) {
        const double* row = storage_h + ((size_t)2 * i + r) * SH;
        const double ji = J_inc[2 * (i - b) + r] + row[12] * dp[0] + row[13] * dp[1] +
                          row[14] * dp[2] + row[15] * dp[3]; /* LMB:617 */
        s += ji * (0.5 * ji + row[16]);                      /* LMB:619 */
      }
    l_diff -= s;
    for (int q = 0; q < 4; ++q)
      lms_h[4 * (size_t)l + q] += dp[q] * jl_col_scale_h[4 * (size_t)l + q]; /* LMB:621-622 */
    free(J_inc);
  }
  free(Nc);
  free(pinc);
  return l_diff;
}

/* apply_joint camera update LZR:283-305 (K13') */
void orc_apply_cam_inc_joint(int32_t n_cams, double* cams, const double* inc11,
                             const double* scaling) {
  for (int c = 0; c < n_cams; ++c) {
    double N[132];
    orc_kernel_basis(12, cams + 12 * (size_t)c, N); /* LZR:300 */
    double d[12];
    for (int k = 0; k < 12; ++k) {
      double s = 0;
      for (int j = 0; j < 11; ++j) s += N[k * 11 + j] * inc11[11 * (size_t)c + j]; /* LZR:301 */
      d[k] = s * scaling[12 * (size_t)c + k];                                      /* LZR:302 */
    }
    for (int k = 0; k < 12; ++k) cams[12 * (size_t)c + k] += d[k]; /* LZR:304 */
  }
}

/* step-2 renormalisation in the outer loop BBA:700-705 (K15) */
void orc_normalize_joint(int32_t n_cams, int32_t n_lms, double* cams, double* lms_h) {
  for (int c = 0; c < n_cams; ++c) {
    double s = 0;
    for (int k = 0; k < 12; ++k) s += cams[12 * (size_t)c + k] * cams[12 * (size_t)c + k];
    s = sqrt(s);
    for (int k = 0; k < 12; ++k) cams[12 * (size_t)c + k] /= s;
  }
  for (int l = 0; l < n_lms; ++l) {
    const double w = lms_h[4 * (size_t)l + 3];
    for (int k = 0; k < 4; ++k) lms_h[4 * (size_t)l + k] /= w;
  }
}

/* ------------------------------------------------------------------------- */
/* explicit Schur complement: LinearizorSC (solver/linearizor_sc.cpp) with     */
/* --solver-type-step-1 PCG | CHOLESKY and --solver-type-step-2 RIPCG         */
/*   LSC = sc/linearization_sc.hpp   CG = cg/conjugate_gradient.hpp            */
/*   PRE = cg/preconditioner.hpp     BSM = cg/block_sparse_matrix.hpp          */
/* The reference keeps S in a hash map of dim x dim blocks (BSM:66-69) and sums */
/* them in hash/TBB order; here S is one dense row-major n x n matrix          */
/* (n = dim * n_cams), absent blocks are zero.                                  */
/* ------------------------------------------------------------------------- */

/* shared body of add_Hb_pOSE LMB:360-412 / add_Hb_joint LMB:414-472 for one landmark.
 * jp(i, r): row r of observation i's pose tile (dim columns), jl likewise (3 columns) */
static void add_hb_landmark(const orc_problem* p, int l, int rows, int dim, const double* jp_base,
                            int jp_stride, const double* jl_base, int jl_stride,
                            const double* res_base, int res_stride, double lambda_lm, double* S,
                            double* b) {
  const size_t n = (size_t)dim * (size_t)p->n_cams;
  const int b0 = p->lm_off[l], e0 = p->lm_off[l + 1];
  double H[9] = {0}, g[3] = {0}, Hi[9], hb[3];
  for (int64_t r = (int64_t)rows * b0; r < (int64_t)rows * e0; ++r) {
    const double* jl = jl_base + r * jl_stride;
    for (int a = 0; a < 3; ++a) {
      for (int c = 0; c < 3; ++c) H[a * 3 + c] += jl[a] * jl[c]; /* LMB:370, 430 */
      g[a] += jl[a] * res_base[r * res_stride];                   /* LMB:372, 433 */
    }
  }
  H[0] += lambda_lm; H[4] += lambda_lm; H[8] += lambda_lm; /* LMB:431 (joint only): Proj^T lambda Proj == lambda I_3 */
  inv3(H, Hi);                                              /* LMB:371, 432 */
  for (int a = 0; a < 3; ++a) hb[a] = Hi[a * 3] * g[0] + Hi[a * 3 + 1] * g[1] + Hi[a * 3 + 2] * g[2];
  double* jltjp = (double*)malloc(sizeof(double) * 3 * (size_t)dim); /* jl_j^T jp_j */
  double* t = (double*)malloc(sizeof(double) * 3 * (size_t)dim);     /* H_ll_inv * (jl_j^T jp_j) */
  double* u = (double*)malloc(sizeof(double) * (size_t)rows * (size_t)dim); /* jl_i * t */
  for (int i = b0; i < e0; ++i) {
    const size_t ci = (size_t)p->cam_idx[i];
    const double* jp_i = jp_base + (size_t)rows * i * jp_stride;
    const double* jl_i = jl_base + (size_t)rows * i * jl_stride;
    for (int a = 0; a < dim; ++a)
      for (int c = 0; c < dim; ++c) {
        double s = 0;
        for (int r = 0; r < rows; ++r) s += jp_i[r * jp_stride + a] * jp_i[r * jp_stride + c];
        S[(ci * dim + a) * n + ci * dim + c] += s; /* LMB:381-384, 445-448 */
      }
    for (int j = b0; j < e0; ++j) {
      const size_t cj = (size_t)p->cam_idx[j];
      const double* jp_j = jp_base + (size_t)rows * j * jp_stride;
      const double* jl_j = jl_base + (size_t)rows * j * jl_stride;
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < dim; ++c) {
          double s = 0;
          for (int r = 0; r < rows; ++r) s += jl_j[r * jl_stride + a] * jp_j[r * jp_stride + c];
          jltjp[a * dim + c] = s;
        }
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < dim; ++c)
          t[a * dim + c] = Hi[a * 3] * jltjp[c] + Hi[a * 3 + 1] * jltjp[dim + c] + Hi[a * 3 + 2] * jltjp[2 * dim + c];
      for (int r = 0; r < rows; ++r)
        for (int c = 0; c < dim; ++c)
          u[r * dim + c] = jl_i[r * jl_stride] * t[c] + jl_i[r * jl_stride + 1] * t[dim + c] +
                           jl_i[r * jl_stride + 2] * t[2 * dim + c];
      for (int a = 0; a < dim; ++a)
        for (int c = 0; c < dim; ++c) {
          double s = 0;
          for (int r = 0; r < rows; ++r) s += -jp_i[r * jp_stride + a] * u[r * dim + c];
          S[(ci * dim + a) * n + cj * dim + c] += s; /* LMB:393-399, 457-463 */
        }
    }
    for (int a = 0; a < dim; ++a) {
      double s = 0;
      for (int r = 0; r < rows; ++r) {
        const double e = res_base[((size_t)rows * i + r) * res_stride] -
                         (jl_i[r * jl_stride] * hb[0] + jl_i[r * jl_stride + 1] * hb[1] + jl_i[r * jl_stride + 2] * hb[2]);
        s += jp_i[r * jp_stride + a] * e;
      }
      b[ci * dim + a] += s; /* LMB:405-408, 468-470 */
    }
  }
  free(jltjp);
  free(t);
  free(u);
}

/* get_Hb_pOSE LSC:403-406 -> get_hb_f_pOSE LSC:462-482 -> add_Hb_pOSE LMB:360-412; pose damping
 * lambda * I added to the diagonal blocks (LSC:477-481).  No landmark damping on this path. */
void orc_get_hb_pose(const orc_problem* p, const double* storage, double lambda_pose, double* S,
                     double* b) {
  const size_t n = 12 * (size_t)p->n_cams;
  memset(S, 0, sizeof(double) * n * n);
  memset(b, 0, sizeof(double) * n); /* LSC:466 */
  for (int l = 0; l < p->n_lms; ++l)
    add_hb_landmark(p, l, 4, 12, storage, ST, storage + 12, ST, storage + 15, ST, 0.0, S, b);
  if (lambda_pose > 0) /* has_pose_damping_pOSE LSC:401, 477 */
    for (size_t k = 0; k < n; ++k) S[k * n + k] += lambda_pose;
}

/* get_Hb_joint LSC:408-411 -> get_hb_f_joint LSC:484-531 -> add_Hb_joint LMB:414-472; pose damping
 * Proj_pose^T lambda Proj_pose == lambda I_11 (LSC:521), landmark damping lambda (LMB:431). */
void orc_get_hb_joint(const orc_problem* p, const double* storage_h, const double* storage_n,
                      double lambda, double* S, double* b) {
  const size_t n = 11 * (size_t)p->n_cams;
  memset(S, 0, sizeof(double) * n * n);
  memset(b, 0, sizeof(double) * n);
  for (int l = 0; l < p->n_lms; ++l)
    add_hb_landmark(p, l, 2, 11, storage_n, SN, storage_n + 11, SN, storage_h + 16, SH, lambda, S, b);
  for (size_t k = 0; k < n; ++k) S[k * n + k] += lambda;
}

/* BlockDiagonalPreconditioner PRE:66-118 with diagonal == nullptr (SCHUR_JACOBI on the explicit S,
 * linearizor_sc.cpp:133-135): inverse of each dim x dim diagonal block by LLT of its upper triangle */
void orc_block_jacobi_inverse(int32_t n_cams, int32_t dim, const double* S, double* inv_blocks) {
  const size_t n = (size_t)dim * (size_t)n_cams;
  for (int c = 0; c < n_cams; ++c) {
    double* B = inv_blocks + (size_t)dim * dim * c;
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j) B[i * dim + j] = S[((size_t)c * dim + i) * n + (size_t)c * dim + j];
    llt_inverse_upper(dim, B); /* PRE:104-107 */
  }
}

static void dense_mul(size_t n, const double* S, const double* x, double* y) { /* BSM right_multiply */
  for (size_t i = 0; i < n; ++i) {
    double s = 0;
    for (size_t j = 0; j < n; ++j) s += S[i * n + j] * x[j];
    y[i] = s;
  }
}

static double dotn(const double* a, const double* b, size_t n) {
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

static int zero_or_inf(double x) { return x == 0.0 || isinf(x); } /* cg/utils.hpp is_zero_or_infinity */

/* ConjugateGradientsSolver::solve CG:112-290 (solve_joint CG:292-470 is the same loop) driven as
 * LinearizorBase::pcg does (linearizor_base.cpp:104-125): r_tolerance = -1, q_tolerance = eta,
 * x starts at zero and is NEGATED on return ("we solve H(-x) = b").  Returns the termination type. */
int orc_pcg(int32_t n_cams, int32_t dim, const double* S, const double* b, const double* inv_blocks,
            int32_t min_iterations, int32_t max_iterations, double eta, double* x,
            int32_t* num_iterations) {
  const size_t n = (size_t)dim * (size_t)n_cams;
  const int residual_reset_period = 10; /* CG:87 */
  int status = ORC_NO_CONVERGENCE, it = 0;
  memset(x, 0, sizeof(double) * n); /* linearizor_sc.cpp:113 */
  const double norm_b = norm2(b, n);
  if (norm_b == 0.0) { /* CG:131-136 */
    *num_iterations = 0;
    return ORC_SUCCESS;
  }
  double* r = (double*)malloc(sizeof(double) * n);
  double* pp = (double*)malloc(sizeof(double) * n);
  double* z = (double*)malloc(sizeof(double) * n);
  double* tmp = (double*)malloc(sizeof(double) * n);
  double* q = (double*)malloc(sizeof(double) * n);
  const double tol_r = -1.0 * norm_b; /* CG:144 with r_tolerance = -1 */
  dense_mul(n, S, x, tmp);
  for (size_t i = 0; i < n; ++i) r[i] = b[i] - tmp[i]; /* CG:146-147 */
  double norm_r = norm2(r, n);
  if (min_iterations == 0 && norm_r <= tol_r) { /* CG:149-158 (never true for tol_r < 0) */
    status = ORC_SUCCESS;
    goto done;
  }
  double rho = 1.0;
  double q0;
  {
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += x[i] * (b[i] + r[i]);
    q0 = -1.0 * s; /* CG:163 */
  }
  for (it = 1;; ++it) {
    if (inv_blocks) orc_right_mul_b_inv(n_cams, dim, inv_blocks, r, z); /* CG:168-172, PRE:120-135 */
    else memcpy(z, r, sizeof(double) * n);
    const double last_rho = rho;
    rho = dotn(r, z, n); /* CG:176 */
    if (zero_or_inf(rho)) { status = ORC_FAILURE; break; } /* CG:177-185 */
    if (it == 1) memcpy(pp, z, sizeof(double) * n);
    else {
      const double beta = rho / last_rho;
      if (zero_or_inf(beta)) { status = ORC_FAILURE; break; } /* CG:191-197 */
      for (size_t i = 0; i < n; ++i) pp[i] = z[i] + beta * pp[i];
    }
    dense_mul(n, S, pp, q); /* CG:201 */
    const double pq = dotn(pp, q, n);
    if (pq <= 0 || isinf(pq)) { status = ORC_NO_CONVERGENCE; break; } /* CG:204-215 */
    const double alpha = rho / pq;
    if (isinf(alpha)) { status = ORC_FAILURE; break; } /* CG:218-223 */
    for (size_t i = 0; i < n; ++i) x[i] = x[i] + alpha * pp[i]; /* CG:225 */
    if (it % residual_reset_period == 0) { /* CG:234-239 */
      dense_mul(n, S, x, tmp);
      for (size_t i = 0; i < n; ++i) r[i] = b[i] - tmp[i];
    } else {
      for (size_t i = 0; i < n; ++i) r[i] = r[i] - alpha * q[i];
    }
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += x[i] * (b[i] + r[i]);
    const double q1 = -1.0 * s; /* CG:243 */
    const double zeta = it * (q1 - q0) / q1; /* CG:268 */
    if (zeta < eta && it >= min_iterations) { status = ORC_SUCCESS; break; } /* CG:269-282 */
    q0 = q1;
    norm_r = norm2(r, n);
    if (norm_r <= tol_r && it >= min_iterations) { status = ORC_SUCCESS; break; } /* CG:288-297 */
    if (it >= max_iterations) break; /* CG:299-301 */
  }
done:
  for (size_t i = 0; i < n; ++i) x[i] = -x[i]; /* linearizor_base.cpp:122 */
  *num_iterations = it;
  free(r); free(pp); free(z); free(tmp); free(q);
  return status;
}

/* solve_direct_pOSE LSC:236-245: accum = SimplicialLLT(H).solve(-b).  Restated with a dense
 * Cholesky (lower factor, no fill-reducing permutation): the exact solve up to rounding.
 * Returns 0, or 1 when the matrix is not positive definite. */
int orc_cholesky_solve(int32_t n_, const double* S, const double* b, double* x) {
  const size_t n = (size_t)n_;
  double* L = (double*)malloc(sizeof(double) * n * n);
  int bad = 0;
  for (size_t j = 0; j < n && !bad; ++j) {
    double d = S[j * n + j];
    for (size_t k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0)) { bad = 1; break; }
    d = sqrt(d);
    L[j * n + j] = d;
    for (size_t i = j + 1; i < n; ++i) {
      double s = S[i * n + j];
      for (size_t k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s / d;
    }
  }
  if (!bad) {
    for (size_t i = 0; i < n; ++i) {
      double s = -b[i];
      for (size_t k = 0; k < i; ++k) s -= L[i * n + k] * x[k];
      x[i] = s / L[i * n + i];
    }
    for (size_t i = n; i-- > 0;) {
      double s = x[i];
      for (size_t k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
      x[i] = s / L[i * n + i];
    }
  }
  free(L);
  return bad;
}
