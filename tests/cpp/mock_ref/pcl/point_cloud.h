// DECLARATION-ONLY mock (see ../README.md)
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
namespace pcl {
template <class PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  void resize(std::size_t);
  std::size_t size() const;
  const PointT& at(std::size_t) const;
};
}  // namespace pcl
