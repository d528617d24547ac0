// rot_vgicp_hip.hpp — header-only C++ drop-in for `fast_gicp::RotVGICP<PointSource, PointTarget>` over the C ABI
// of librolo_hip.so (include/rolo_hip.h). It keeps the reference's class name, method names, argument meaning
// and error behaviour (reference include/rot_gicp/gicp/rot_vgicp.hpp:72-104, lsq_registration.hpp:51-62), so
// src/lidarOdometry.cpp:460-494 compiles against it unchanged:
//
//     #include <rot_vgicp_hip.hpp>               // instead of <rot_gicp/gicp/rot_vgicp.hpp>
//     fast_gicp::RotVGICP<PointType, PointType> rot_vgicp;
//     rot_vgicp.setPolarResolution(0.175, 0.175, 2.0);
//     rot_vgicp.setInputTarget(featureLast); rot_vgicp.setInputSource(feature_propagated);
//     rot_vgicp.align(*aligned);
//     Eigen::Matrix4f trans = rot_vgicp.getFinalTransformation();
//     rot_vgicp.computeTranslation(*aligned, Reg_translation, Translation, TranslationOld, 0.1, 0.1, CT_lambda);
//
// Two build modes:
//   * with PCL + Eigen on the include path (the ROLO catkin workspace): define ROLO_HIP_WITH_PCL before including;
//     the class then takes pcl::PointCloud<PointT>::ConstPtr / Eigen types exactly like the reference (type-checked here against
//     declaration-level stand-ins of those types: tests/cpp/shim_pcl_check.cpp, tests/cpp/mock_pcl);
//   * without them (this repository's CI image has neither): a POD cloud `rolo::Cloud` (n x 8 floats, the
//     pcl::PointXYZI memory layout) and plain arrays stand in, same methods.
// The class owns one rolo_ctx (one HIP stream); like the reference object it is not thread-safe.
#pragma once
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rolo_hip.h"

#ifdef ROLO_HIP_WITH_PCL
#include <Eigen/Core>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#endif

namespace rolo {

// pcl::PointXYZI memory layout: float data[4] (x, y, z, 1), float intensity, 12 bytes padding
struct PointXYZI { float x, y, z, w, intensity, pad[3]; };
static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes");
struct Cloud {
  std::vector<PointXYZI> points;
  size_t size() const { return points.size(); }
  using Ptr = std::shared_ptr<Cloud>;
  using ConstPtr = std::shared_ptr<const Cloud>;
};

}  // namespace rolo

namespace fast_gicp {

// gicp_settings.hpp:6-13, lsq_registration.hpp:13 — same enumerators, same order
enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS, PLANE_S };
enum class NeighborSearchMethod { DIRECT27, DIRECT7, DIRECT1, DIRECT_RADIUS };
enum class VoxelAccumulationMode { ADDITIVE, ADDITIVE_WEIGHTED, MULTIPLICATIVE };
enum class VoxelType { POLAR, UNIFORM };
enum class LSQ_OPTIMIZER_TYPE { GaussNewton, LevenbergMarquardt, SO3_LevenbergMarquardt };

#ifdef ROLO_HIP_WITH_PCL
template <typename PointSource, typename PointTarget>
#else
template <typename PointSource = rolo::PointXYZI, typename PointTarget = rolo::PointXYZI>
#endif
class RotVGICP {
public:
#ifdef ROLO_HIP_WITH_PCL
  using PointCloudSource = pcl::PointCloud<PointSource>;
  using PointCloudTarget = pcl::PointCloud<PointTarget>;
  using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
  using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
  using Matrix4 = Eigen::Matrix4f;
  using Vector3d = Eigen::Vector3d;
  using Matrix4d = Eigen::Matrix4d;
  using CovarianceList = std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>>;
  using Matrix6d = Eigen::Matrix<double, 6, 6>;
  using Vector6d = Eigen::Matrix<double, 6, 1>;
#else
  using PointCloudSource = rolo::Cloud;
  using PointCloudTarget = rolo::Cloud;
  using PointCloudSourceConstPtr = rolo::Cloud::ConstPtr;
  using PointCloudTargetConstPtr = rolo::Cloud::ConstPtr;
  using Matrix4 = std::array<float, 16>;   // row-major
  using Vector3d = std::array<double, 3>;
  using Matrix4d = std::array<double, 16>;  // row-major
  using CovarianceList = std::vector<Matrix4d>;
  using Matrix6d = std::array<double, 36>;  // row-major
  using Vector6d = std::array<double, 6>;
#endif
  static_assert(sizeof(PointSource) % sizeof(float) == 0 && sizeof(PointTarget) % sizeof(float) == 0, "points must be float records");

  // The reference constructs this object once per frame (src/lidarOdometry.cpp:460): construction / destruction go through the library's
  // context pool (rolo_ctx_acquire / _release) — a released context keeps its streams and device buffers and comes back reset to a fresh
  // object's state, so the per-frame object costs no hipMalloc / hipFree.
  // This object registers ONE frame at a time (align, then computeTranslation, on the caller's thread): it asks for the latency form of the LM chain —
  // one launch per trial (rolo_params.fused_lm; the library default is the throughput form, which pays off only with several contexts in flight).
  explicit RotVGICP(int device = 0) {
    if (rolo_ctx_acquire(device, &ctx_) != ROLO_OK) throw std::runtime_error(std::string("RotVGICP(HIP): ") + rolo_last_error());
    rolo_default_params(&p_);
    p_.fused_lm = 1;
    for (int i = 0; i < 16; i++) final_[i] = (i % 5 == 0) ? 1.f : 0.f;
    push();
  }
  ~RotVGICP() { rolo_ctx_release(ctx_); }
  RotVGICP(const RotVGICP&) = delete;
  RotVGICP& operator=(const RotVGICP&) = delete;

  // ---- rot_vgicp_impl.hpp:44-110 ----
  void setResolution(double resolution) { p_.voxel_resolution = resolution; p_.voxel_type = ROLO_VOXEL_UNIFORM; push(); }
  void setPolarResolution(double theta_res, double phi_res, double r_res) {
    p_.polar_resolution[0] = theta_res; p_.polar_resolution[1] = phi_res; p_.polar_resolution[2] = r_res;
    p_.voxel_type = ROLO_VOXEL_POLAR; push();
  }
  void setNeighborSearchMethod(NeighborSearchMethod m) {
    if (m == NeighborSearchMethod::DIRECT_RADIUS) { std::fprintf(stderr, "unsupported neighbor search method\n"); std::abort(); }  // vmp_voxel.hpp:16-18
    p_.neighbor_search = static_cast<int>(m); push();
  }
  void setVoxelAccumulationMode(VoxelAccumulationMode) {}  // every mode builds AdditiveVmfVoxel (vmp_voxel.hpp:176-184)
  void setNumThreads(int) {}                               // no host threads on the HIP path
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; push(); }
  void setRegularizationMethod(RegularizationMethod m) { p_.regularization = static_cast<int>(m); push(); }
  void setOptimizerType(LSQ_OPTIMIZER_TYPE t) { p_.optimizer = static_cast<int>(t); push(); }
  void setRotationEpsilon(double eps) { p_.rotation_epsilon = eps; push(); }
  void setTransformationEpsilon(double eps) { p_.transformation_epsilon = eps; push(); }
  void setMaximumIterations(int n) { p_.max_iterations = n; push(); }
  void setInitialLambdaFactor(double f) { p_.lm_init_lambda_factor = f; push(); }
  // lm_debug_print_ (lsq_registration.hpp:60): the per-trial table of rot_step_lm / step_lm / step_t_optimize (lsq_registration_impl.hpp:114-120, :245-251,
  // :299-305) and computeTransformation's banner (:158-162), printed to stdout in the reference's boost::format layout — from the device-side LM trace
  // (rolo_get_trace) after the solve, since the trials themselves run on the GPU without a host round trip
  void setDebugPrint(bool lm_debug_print) { lm_debug_print_ = lm_debug_print; }

  void clearSource() { src_.reset(); check(rolo_clear_source(ctx_)); }
  void clearTarget() { tgt_.reset(); check(rolo_clear_target(ctx_)); }
  void swapSourceAndTarget() { src_.swap(tgt_); check(rolo_swap_source_and_target(ctx_)); }

  void setInputSource(const PointCloudSourceConstPtr& cloud) {
    if (src_ == cloud) return;  // :113-115
    src_ = cloud;
    check(rolo_set_source(ctx_, reinterpret_cast<const float*>(cloud->points.data()), (int)cloud->points.size(), (int)(sizeof(PointSource) / sizeof(float))));
  }
  void setInputTarget(const PointCloudTargetConstPtr& cloud) {
    if (tgt_ == cloud) return;  // :134-136
    tgt_ = cloud;
    check(rolo_set_target(ctx_, reinterpret_cast<const float*>(cloud->points.data()), (int)cloud->points.size(), (int)(sizeof(PointTarget) / sizeof(float))));
  }

  // pcl::Registration::align(output) -> computeTransformation (:146-160)
  void align(PointCloudSource& output) { Matrix4 guess = identity(); align(output, guess); }
  void align(PointCloudSource& output, const Matrix4& guess) {
    if (!src_ || !tgt_) throw std::runtime_error("RotVGICP: source / target not set");
    if (output.points.data() == src_->points.data() || output.points.data() == tgt_->points.data())
      throw std::invalid_argument("RotVGICP: destination cloud cannot be identical to source or target");  // :150-152
    float g[16]; to_rowmajor(guess, g);
    rolo_stats st;
    const int rc = rolo_align(ctx_, g, final_, nullptr, &st);
    if (rc != ROLO_OK) throw std::runtime_error(std::string("RotVGICP(HIP) align: ") + rolo_last_error());
    if (st.lm_failed) std::fprintf(stderr, "lm not converged!!\n");  // lsq_registration_impl.hpp:168-171
    converged_ = st.converged != 0; if (st.n_outer > 0) nr_iterations_ = st.n_outer - 1;   // (:163 — a loop that never runs leaves the member as it was)
    if (lm_debug_print_) {
      std::printf("********************************************\n***************** optimize *****************\n********************************************\n");
      print_trace(0);
    }
    transform_into(output, final_);  // lsq_registration_impl.hpp:178
  }
  Matrix4 getFinalTransformation() const { return from_rowmajor(final_); }
  bool hasConverged() const { return converged_; }

  // RotVGICP::computeTranslation (:163-169)
  void computeTranslation(PointCloudSource& output, Vector3d& trans, const Vector3d& init_guess, const Vector3d& last_t0,
                          const double interval_tn, const double interval_tn_1, const float ct_lambda) {
    double t[3] = {trans[0], trans[1], trans[2]}, g[3] = {init_guess[0], init_guess[1], init_guess[2]}, l[3] = {last_t0[0], last_t0[1], last_t0[2]};
    rolo_stats st;
    const int rc = rolo_compute_translation(ctx_, t, g, l, interval_tn, interval_tn_1, ct_lambda, &st);
    if (rc != ROLO_OK) throw std::runtime_error(std::string("RotVGICP(HIP) computeTranslation: ") + rolo_last_error());
    if (st.lm_failed) std::fprintf(stderr, "lm not converged!!\n");
    if (lm_debug_print_) print_trace(1);
    trans[0] = t[0]; trans[1] = t[1]; trans[2] = t[2];
    float T[16] = {1, 0, 0, (float)t[0], 0, 1, 0, (float)t[1], 0, 0, 1, (float)t[2], 0, 0, 0, 1};
    transform_into(output, T);  // lsq_registration_impl.hpp:75-78
  }

  // ---- rot_vgicp.hpp:89-97: covariances in, covariances out (Matrix4d per point, last row / column zero) ----
  void setSourceCovariances(const CovarianceList& covs) {   // rot_vgicp_impl.hpp:122-125
    if (!src_ || covs.size() != src_->points.size()) throw std::invalid_argument("RotVGICP: one covariance per source point");
    std::vector<double> m; flatten(covs, m);
    check(rolo_set_source_covariances(ctx_, m.data()));
  }
  void setTargetCovariances(const CovarianceList& covs) {   // :127-130
    if (!tgt_ || covs.size() != tgt_->points.size()) throw std::invalid_argument("RotVGICP: one covariance per target point");
    std::vector<double> m; flatten(covs, m);
    check(rolo_set_target_covariances(ctx_, m.data()));
  }
  // the reference returns its member vector (empty before the first align); here the covariances live in HBM and are fetched,
  // computed first if a cloud has none yet
  const CovarianceList& getSourceCovariances() { fetch_covs(true); return source_covs_; }
  const CovarianceList& getTargetCovariances() { fetch_covs(false); return target_covs_; }

  // ---- lsq_registration.hpp:55-57 ----
  const Matrix6d& getFinalHessian() {   // lsq_registration_impl.hpp:45-47: Identity until a 6-dof LM step accepts
    double H[36];
    check(rolo_get_final_hessian(ctx_, H));
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) at6(final_hessian_, i, j) = H[i * 6 + j];
    return final_hessian_;
  }
  // evaluateCost(relative_pose, H, b) = linearize(Isometry3f(relative_pose).cast<double>(), H, b)  (:50-52): the 6-dof linearisation
  double evaluateCost(const Matrix4& relative_pose, Matrix6d* H = nullptr, Vector6d* b = nullptr) {
    float g[16]; to_rowmajor(relative_pose, g);
    double T[16], H36[36], b6[6], err = 0;
    for (int i = 0; i < 16; i++) T[i] = (double)g[i];
    check(rolo_linearize(ctx_, T, H36, b6, &err));
    if (H) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) at6(*H, i, j) = H36[i * 6 + j];
    if (b) for (int i = 0; i < 6; i++) (*b)[i] = b6[i];
    return err;
  }

  rolo_ctx* handle() { return ctx_; }

protected:
  // lsq_registration.hpp:104: lm_max_iterations_ is a protected member without a setter in the reference — what a subclass can change there, a subclass can change here
  void set_lm_max_iterations(int n) { p_.lm_max_iterations = n; push(); }

private:
  // the trials of one stage (0 = align, 1 = computeTranslation) in the reference's table: a header line before every trial 0, then
  // boost::format("%5d %15g %15g %15g %15g %15g %5c") % i % y0 % yi % rho % lm_lambda_ % d.norm() % dec — identical to printf's conversions
  void print_trace(int stage) {
    const int n = rolo_get_trace(ctx_, nullptr, 0);
    if (n <= 0) return;
    std::vector<rolo_trace_rec> tr((size_t)n);
    if (rolo_get_trace(ctx_, tr.data(), n) < 0) return;
    for (const rolo_trace_rec& r : tr) {
      if (r.stage != stage || r.yi != r.yi) continue;   // (Gauss-Newton steps carry NaN: step_gn prints nothing)
      if (r.trial == 0) std::printf("--- LM optimization ---\n%5s %15s %15s %15s %15s %15s %5s\n", "i", "y0", "yi", "rho", "lambda", "|delta|", "dec");
      std::printf("%5d %15g %15g %15g %15g %15g %5c\n", r.trial, r.y0, r.yi, r.rho, r.lambda, r.dnorm, r.rho > 0.0 ? 'x' : ' ');
    }
    std::fflush(stdout);
  }
  bool lm_debug_print_ = false;
  void push() { check(rolo_set_params(ctx_, &p_)); }
  static void check(int rc) { if (rc != ROLO_OK) throw std::runtime_error(std::string("RotVGICP(HIP): ") + rolo_last_error()); }
  // pcl::transformPointCloud(*input_, output, T) (lsq_registration_impl.hpp:78, :178) ON THE HOST, as the reference does it: the whole cloud copied (header,
  // width / height, every field), then x, y, z replaced by the scalar path's  c0 x + c1 y + c2 z + c3  (left to right) and w by 1. Until round 4 this went
  // through rolo_transform_cloud — the source uploaded a second time, a kernel, a 1 MB download and a host wait, twice per frame (~0.2 of the 0.74 ms a
  // VLP-16 frame took through this class) — for a cloud the reference caller never reads (src/lidarOdometry.cpp:468, :491: `aligned` is cleared and dropped).
  // The sums must stay four separate float operations per coordinate (multiply, add, add, add): that is what rolo_transform_cloud and the oracle compute, and what the
  // bit-equality of `aligned` with them rests on. A caller's compiler is free to contract a * x + b * y into an FMA (GCC defaults to -ffp-contract=fast in gnu++
  // modes once -march=native / -mfma makes FMA available), so contraction is switched off for this one function whatever the translation unit is built with.
#if defined(__clang__)
#define ROLO_HIP_NO_FP_CONTRACT _Pragma("clang fp contract(off)")
#define ROLO_HIP_NO_FP_CONTRACT_ATTR
#elif defined(__GNUC__)
#define ROLO_HIP_NO_FP_CONTRACT
#define ROLO_HIP_NO_FP_CONTRACT_ATTR __attribute__((optimize("fp-contract=off")))
#else
#define ROLO_HIP_NO_FP_CONTRACT
#define ROLO_HIP_NO_FP_CONTRACT_ATTR
#endif
  ROLO_HIP_NO_FP_CONTRACT_ATTR void transform_into(PointCloudSource& output, const float* T) {
    ROLO_HIP_NO_FP_CONTRACT
    output = *src_;
    constexpr size_t stride = sizeof(PointSource) / sizeof(float);
    float* p = output.points.empty() ? nullptr : reinterpret_cast<float*>(output.points.data());
    const size_t n = output.points.size();
    for (size_t i = 0; i < n; i++, p += stride) {
      const float x = p[0], y = p[1], z = p[2];
      p[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
      p[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
      p[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
      p[3] = 1.0f;
    }
  }
  void fetch_covs(bool source) {
    const size_t n = source ? (src_ ? src_->points.size() : 0) : (tgt_ ? tgt_->points.size() : 0);
    CovarianceList& out = source ? source_covs_ : target_covs_;
    out.resize(n);
    if (!n) return;
    check(rolo_compute_covariances(ctx_));   // no-op for clouds that hold covariances already
    std::vector<double> m(16 * n);
    check(source ? rolo_get_source_covariances(ctx_, m.data()) : rolo_get_target_covariances(ctx_, m.data()));
    for (size_t k = 0; k < n; k++) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) at4(out[k], i, j) = m[16 * k + 4 * i + j];
  }
  static void flatten(const CovarianceList& covs, std::vector<double>& m) {
    m.resize(16 * covs.size());
    for (size_t k = 0; k < covs.size(); k++) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m[16 * k + 4 * i + j] = at4c(covs[k], i, j);
  }
#ifdef ROLO_HIP_WITH_PCL
  static double& at4(Matrix4d& m, int i, int j) { return m(i, j); }
  static double at4c(const Matrix4d& m, int i, int j) { return m(i, j); }
  static double& at6(Matrix6d& m, int i, int j) { return m(i, j); }
#else
  static double& at4(Matrix4d& m, int i, int j) { return m[i * 4 + j]; }
  static double at4c(const Matrix4d& m, int i, int j) { return m[i * 4 + j]; }
  static double& at6(Matrix6d& m, int i, int j) { return m[i * 6 + j]; }
#endif
#ifdef ROLO_HIP_WITH_PCL
  static Matrix4 identity() { return Matrix4::Identity(); }
  static void to_rowmajor(const Matrix4& m, float* o) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) o[i * 4 + j] = m(i, j); }
  static Matrix4 from_rowmajor(const float* o) { Matrix4 m; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m(i, j) = o[i * 4 + j]; return m; }
#else
  static Matrix4 identity() { Matrix4 m{}; for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.f : 0.f; return m; }
  static void to_rowmajor(const Matrix4& m, float* o) { std::memcpy(o, m.data(), sizeof(float) * 16); }
  static Matrix4 from_rowmajor(const float* o) { Matrix4 m; std::memcpy(m.data(), o, sizeof(float) * 16); return m; }
#endif

  rolo_ctx* ctx_ = nullptr;
  rolo_params p_;
  PointCloudSourceConstPtr src_;
  PointCloudTargetConstPtr tgt_;
  float final_[16];
  bool converged_ = false;
  int nr_iterations_ = 0;
  CovarianceList source_covs_, target_covs_;
  Matrix6d final_hessian_{};
};

}  // namespace fast_gicp
