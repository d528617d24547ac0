// DECLARATION-ONLY mock (see ../README.md)
#pragma once
#include <Eigen/Core>
namespace pcl {
struct PointXYZI {
  float x, y, z, data3, intensity, pad[3];
  Eigen::CastProxy<float, 4, 1> getVector4fMap() const;   // Eigen::Map<const Vector4f>: the tool only calls .cast<double>() on it
};
}  // namespace pcl
