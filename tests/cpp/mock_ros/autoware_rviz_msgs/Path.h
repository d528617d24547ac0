// autoware_rviz_msgs (an external package the reference depends on, package.xml / CMakeLists.txt): the fields src/lidarOdometry.cpp:277-316 fills
#pragma once
#include <geometry_msgs/PoseStamped.h>
#include <vector>
namespace autoware_rviz_msgs {
struct PathPoint { geometry_msgs::Pose pose; double longitudinal_velocity_mps = 0, lateral_velocity_mps = 0, heading_rate_rps = 0; bool is_final = false; };
struct Path { std_msgs::Header header; std::vector<PathPoint> points; };
}  // namespace autoware_rviz_msgs
