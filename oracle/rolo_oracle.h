/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header). PARITY UNPINNED by the reference itself.
 *
 * C ABI of the CPU restatement of sdwyc/ROLO's per-frame scan-matching hot path. Citations are relative to
 * /root/reference. Used by tests/ (as the checker), __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#ifndef ROLO_ORACLE_H
#define ROLO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* enum orders follow include/rot_gicp/gicp/gicp_settings.hpp:6-13 and lsq_registration.hpp:13 */
enum { ORC_REG_NONE = 0, ORC_REG_MIN_EIG, ORC_REG_NORMALIZED_MIN_EIG, ORC_REG_PLANE, ORC_REG_FROBENIUS, ORC_REG_PLANE_S };
enum { ORC_DIRECT27 = 0, ORC_DIRECT7, ORC_DIRECT1 };
enum { ORC_VOXEL_POLAR = 0, ORC_VOXEL_UNIFORM };
enum { ORC_OPT_GN = 0, ORC_OPT_LM, ORC_OPT_SO3_LM };

typedef struct orc_params {
  int k_correspondences;         /* rot_vgicp_impl.hpp:28  (20) */
  int regularization;            /* :30 PLANE */
  int neighbor_search;           /* :35 DIRECT1 */
  int voxel_type;                /* :37 POLAR */
  double voxel_resolution;       /* :33 1.0 */
  double polar_resolution[3];    /* :38 (1,0,0) until setPolarResolution; caller sets 0.175,0.175,2.0 (lidarOdometry.cpp:462) */
  int optimizer;                 /* lsq_registration_impl.hpp:16 SO3_LM */
  int max_iterations;            /* :11 64 */
  double rotation_epsilon;       /* :12 2e-3 */
  double transformation_epsilon; /* :13 5e-4 */
  int lm_max_iterations;         /* :18 10 */
  double lm_init_lambda_factor;  /* :19 1e-9 */
  int num_threads;               /* 0 => omp_get_max_threads (rot_vgicp_impl.hpp:81-88) */
  int fixed_iterations;          /* harness knob: >0 forces exactly this many outer iterations of align() */
  int q2_intended;               /* 0 = as written (last_transform keeps its initial value), 1 = intended */
} orc_params;

typedef struct orc_trace_rec {
  int stage;    /* 0 rotation / 6-dof, 1 translation */
  int outer;    /* outer iteration index */
  int trial;    /* LM trial index j */
  int accepted; /* 1 accepted, 0 rejected, 2 rejected-but-converged (returns true without update) */
  double y0, yi, rho, lambda, dnorm;
} orc_trace_rec;

void orc_default_params(orc_params* p);

typedef struct orc_reg orc_reg;
orc_reg* orc_reg_create(const orc_params* p);
void orc_reg_destroy(orc_reg* r);
/* points: n records of `stride` floats, x,y,z at offsets 0,1,2 (PCL PointXYZI: stride 8). Copied. */
int orc_reg_set_target(orc_reg* r, const float* pts, int n, int stride);
int orc_reg_set_source(orc_reg* r, const float* pts, int n, int stride);
/* K5: rot_vgicp_impl.hpp:421-496. covs: n x 16 doubles, row-major 4x4. */
int orc_reg_compute_covariances(orc_reg* r);
int orc_reg_get_source_covs(orc_reg* r, double* covs);
int orc_reg_get_target_covs(orc_reg* r, double* covs);
/* setSourceCovariances (rot_vgicp_impl.hpp:122-125): n x 16 doubles */
int orc_reg_set_source_covs(orc_reg* r, const double* covs);
/* K6: vmp_voxel.hpp:167-197. Builds the map from the target (needs covariances). */
int orc_reg_build_voxelmap(orc_reg* r);
int orc_reg_num_voxels(orc_reg* r);
/* voxel order = order of first appearance in the target cloud. keys V x 3, counts V, means V x 4, covs V x 16 */
int orc_reg_get_voxels(orc_reg* r, int32_t* keys, int32_t* counts, double* means, double* covs);
/* K7/K8/K9/K10: evaluate at a given pose (row-major 4x4 double). H/b may be NULL. */
double orc_reg_so3_linearize(orc_reg* r, const double* T, double* H9, double* b3);
double orc_reg_linearize(orc_reg* r, const double* T, double* H36, double* b6);
double orc_reg_compute_error(orc_reg* r, const double* T);
int orc_reg_num_correspondences(orc_reg* r);
int orc_reg_get_correspondences(orc_reg* r, int32_t* src_idx, int32_t* voxel_idx, double* mahalanobis /* Nc x 16 or NULL */);
/* K11 */
double orc_reg_t3_linearize(orc_reg* r, const double* t3, const double* init_guess3, const double* last_t03,
                            double dtn, double dtn1, float ct_lambda, double* H36, double* b6);
double orc_reg_compute_t_error(orc_reg* r, const double* t3, const double* init_guess3, const double* last_t03,
                               double dtn, double dtn1, float ct_lambda);
/* K12 drivers. align: pcl::Registration::align(out, guess) -> computeTransformation. guess/T_out row-major 4x4.
 * T_out_f is final_transformation_ (float), T_out_d the double x0 it was cast from. Returns 0, or <0 on error. */
int orc_reg_align(orc_reg* r, const float* guess16, float* T_out_f16, double* T_out_d16, int* n_outer, int* converged);
int orc_reg_compute_translation(orc_reg* r, double* trans3_io, const double* init_guess3, const double* last_t03,
                                double dtn, double dtn1, float ct_lambda, int* n_outer);
int orc_reg_trace(orc_reg* r, orc_trace_rec* out, int cap); /* returns number of records (may exceed cap) */
/* Scripted evaluations for tests of the LM drivers' exits (lsq_registration_impl.hpp:55-179, 225-324): while a script is set, every linearisation
 * (so3_linearize / linearize / t3_linearize) opened by outer iteration o returns lin_*[o] and every trial cost (compute_error / compute_t_error) of
 * (o, trial t) returns err_y[o][t]; no clouds are needed. The arrays stay owned by the caller; NULL switches back to real evaluations. */
typedef struct orc_lm_script {
  int n_outer, n_trial;  /* extents; indices beyond them repeat the last entry */
  const double* lin_y;   /* [n_outer] */
  const double* lin_H;   /* [n_outer][36] row-major 6 x 6; a 3-dof optimiser reads the top-left 3 x 3 */
  const double* lin_b;   /* [n_outer][6] */
  const int* lin_n;      /* [n_outer] number of correspondences of that linearisation */
  const double* err_y;   /* [n_outer][n_trial] */
} orc_lm_script;
void orc_reg_set_script(orc_reg* r, const orc_lm_script* s);
/* the setters of the LM drivers' knobs on a live object (setRotationEpsilon, setTransformationEpsilon, setMaximumIterations, setInitialLambdaFactor, setOptimizerType,
 * lm_max_iterations_): everything cached — covariances, map, correspondences — is kept, as the reference's setters keep it */
void orc_reg_set_driver_params(orc_reg* r, const orc_params* p);
void orc_reg_clear_trace(orc_reg* r);

/* stand-alone pieces, for unit parity */
int orc_knn(const float* pts, int n, int stride, int k, int threads, int32_t* idx /* n*k */, float* d2 /* n*k */);
int orc_voxel_keys(const float* pts, int n, int stride, int voxel_type, double voxel_resolution,
                   const double* polar_res3, const double* T /* row-major 4x4 or NULL */, int32_t* keys /* n*3 */);
void orc_so3_exp(const double* omega3, double* R9);
void orc_se3_exp(const double* a6, double* R9, double* t3);
void orc_svd3(const double* A9, double* U9, double* s3, double* V9);
int orc_ldlt_solve(int n /*3 or 6*/, const double* A, const double* rhs, double* x);
/* pcl::transformPointCloud (float path), lidarOdometry.cpp:459,492,581; lsq_registration_impl.hpp:78,178 */
void orc_transform_cloud_f(const float* in, float* out, int n, int stride, const float* T16);

#ifdef __cplusplus
}
#endif
#endif
