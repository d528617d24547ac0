#pragma once
#include <geometry_msgs/PoseStamped.h>
namespace geometry_msgs { struct PoseWithCovarianceStamped { std_msgs::Header header; PoseWithCovariance pose; }; }
