#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <class PointT>
struct PointCloud {
  std::vector<PointT> points;
  size_t size() const { return points.size(); }
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;        // boost::shared_ptr in PCL <= 1.10, std::shared_ptr after
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
};
}  // namespace pcl
