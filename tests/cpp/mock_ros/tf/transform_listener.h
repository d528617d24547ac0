#pragma once
#include <tf/transform_broadcaster.h>
namespace tf {
struct TransformListener {
  bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&) const { return true; }
  void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) const {}
};
}  // namespace tf
