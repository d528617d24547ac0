// DECLARATION-ONLY mock (see ../../README.md) of what tools/dump_reference_golden.cpp touches of the reference's operator:
//   include/rot_gicp/gicp/rot_vgicp.hpp:72-160 (public setters, computeTranslation, the protected stage functions and members),
//   include/rot_gicp/gicp/lsq_registration.hpp:51-124 (LsqRegistration: nr_iterations_/final_transformation_ come from pcl::Registration),
//   include/rot_gicp/gicp/vmp_voxel.hpp:60-233 (VmfVoxel, VmfVoxelMap), include/rot_gicp/gicp/gicp_settings.hpp (VoxelType).
// Nothing here has a body; nothing can be linked or run.
#pragma once
#include <memory>
#include <utility>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace fast_gicp {
enum class VoxelType { POLAR, UNIFORM };
struct VmfVoxel {
  typedef std::shared_ptr<VmfVoxel> Ptr;
  int num_points;
  Eigen::Vector4d mean_dir;
  Eigen::Matrix4d cov;
};
template <class PointT>
struct VmfVoxelMap {
  Eigen::Vector3i polar_coord(const Eigen::Vector4d& x) const;    // vmp_voxel.hpp:203-211
  Eigen::Vector3i voxel_coord(const Eigen::Vector4d& x) const;    // vmp_voxel.hpp:199-201
  VmfVoxel::Ptr lookup_voxel(const Eigen::Vector3i& x) const;     // vmp_voxel.hpp:226-233
};
template <class PointSource, class PointTarget>
class RotVGICP {
public:
  typedef pcl::PointCloud<PointSource> PointCloudSource;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef typename pcl::PointCloud<PointTarget>::ConstPtr PointCloudTargetConstPtr;
  typedef std::vector<Eigen::Matrix4d, Eigen::aligned_allocator<Eigen::Matrix4d>> CovarianceList;
  RotVGICP();
  virtual ~RotVGICP();
  void setPolarResolution(double, double, double);
  void setResolution(double);
  void setNumThreads(int);
  void clearTarget();
  void clearSource();
  void setInputTarget(const PointCloudTargetConstPtr&);
  void setInputSource(const PointCloudSourceConstPtr&);
  const CovarianceList& getSourceCovariances() const;
  const CovarianceList& getTargetCovariances() const;
  void align(PointCloudSource& output);                            // pcl::Registration
  Eigen::Matrix4f getFinalTransformation() const;                 // pcl::Registration
  bool hasConverged() const;                                      // pcl::Registration
  void computeTranslation(PointCloudSource& output, Eigen::Vector3d& trans, const Eigen::Vector3d& init_guess, const Eigen::Vector3d& last_t0,
                          const double interval_tn, const double interval_tn_1, const float ct_lambda);
protected:
  double so3_linearize(const Eigen::Isometry3d& trans, Eigen::Matrix<double, 3, 3>* H, Eigen::Matrix<double, 3, 1>* b);
  double linearize(const Eigen::Isometry3d& trans, Eigen::Matrix<double, 6, 6>* H, Eigen::Matrix<double, 6, 1>* b);
  double compute_error(const Eigen::Isometry3d& trans);
  double t3_linearize(const Eigen::Vector3d& trans, const Eigen::Vector3d& init_guess, const Eigen::Vector3d& last_t0, const double interval_tn,
                      const double interval_tn_1, Eigen::Matrix<double, 6, 6>* H, Eigen::Matrix<double, 6, 1>* b);
  double compute_t_error(const Eigen::Vector3d& trans, const Eigen::Vector3d& init_guess, const Eigen::Vector3d& last_t0, const double interval_tn,
                         const double interval_tn_1);
  std::unique_ptr<VmfVoxelMap<PointTarget>> voxelmap_;
  std::vector<std::pair<int, VmfVoxel::Ptr>> voxel_correspondences_;
  VoxelType voxel_type_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  int nr_iterations_;
  Eigen::Matrix4f final_transformation_;
  float lambda_;
};
}  // namespace fast_gicp
