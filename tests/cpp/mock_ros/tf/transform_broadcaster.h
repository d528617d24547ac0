#pragma once
#include <ros/ros.h>
namespace tf {
struct Quaternion { Quaternion(double, double, double, double) {} };
struct Vector3 { Vector3(double, double, double) {} };
struct Transform { Transform(const Quaternion&, const Vector3&) {} };
struct StampedTransform { StampedTransform(const Transform&, const ros::Time&, const std::string&, const std::string&) {} };
struct TransformBroadcaster { void sendTransform(const StampedTransform&) {} };
}  // namespace tf
