#pragma once
#include <geometry_msgs/PoseStamped.h>
#include <vector>
namespace nav_msgs { struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; }; }
