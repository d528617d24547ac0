#pragma once
#include <std_msgs/Header.h>
#include <array>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance{}; };
struct Twist { Vector3 linear, angular; };
struct TwistWithCovariance { Twist twist; std::array<double, 36> covariance{}; };
}  // namespace geometry_msgs
