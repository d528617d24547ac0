#pragma once
#include <geometry_msgs/PoseStamped.h>
namespace nav_msgs {
struct Odometry { std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; geometry_msgs::TwistWithCovariance twist; };
typedef boost::shared_ptr<Odometry const> OdometryConstPtr;
}  // namespace nav_msgs
