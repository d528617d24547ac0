#pragma once
#include <ros/ros.h>
#include <geometry_msgs/PoseStamped.h>
#include <stdexcept>
namespace tf {
struct Quaternion { Quaternion() {} Quaternion(double, double, double, double) {} };
struct Vector3 { Vector3() {} Vector3(double, double, double) {} };
struct Transform { Transform() {} Transform(const Quaternion&, const Vector3&) {} Transform operator*(const Transform&) const { return Transform(); } };
struct StampedTransform : Transform { StampedTransform() {} StampedTransform(const Transform&, const ros::Time&, const std::string&, const std::string&) {} };
struct TransformBroadcaster { void sendTransform(const StampedTransform&) {} };
struct TransformException : std::runtime_error { explicit TransformException(const std::string& s) : std::runtime_error(s) {} };
inline Quaternion createQuaternionFromRPY(double, double, double) { return Quaternion(); }
inline void poseMsgToTF(const geometry_msgs::Pose&, Transform&) {}
}  // namespace tf
