// Field-by-field copies between the ROS message classes and the wire structs of include/rolo_ros_wire.hpp. Only the catkin node
// sources in this directory include this file (it needs roscpp, sensor_msgs, nav_msgs, geometry_msgs and the generated
// rolo/CloudInfoStamp.h); nothing else in the repository does. Type-checked in this repository against declaration-only stand-ins of the ROS
// API (tests/test_ros_sources_compile.py; no ROS in the image) — the logic the nodes run lives in include/rolo_ros_nodes.hpp.
#pragma once
#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/PoseWithCovarianceStamped.h>
#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "rolo/CloudInfoStamp.h"
#include "rolo_ros_nodes.hpp"

namespace rolo {
namespace ros1 {

inline wire::Header from_ros(const std_msgs::Header& h) { wire::Header o; o.seq = h.seq; o.stamp.sec = h.stamp.sec; o.stamp.nsec = h.stamp.nsec; o.frame_id = h.frame_id; return o; }
inline std_msgs::Header to_ros(const wire::Header& h) { std_msgs::Header o; o.seq = h.seq; o.stamp.sec = h.stamp.sec; o.stamp.nsec = h.stamp.nsec; o.frame_id = h.frame_id; return o; }

inline wire::PointCloud2 from_ros(const sensor_msgs::PointCloud2& m) {
  wire::PointCloud2 o;
  o.header = from_ros(m.header); o.height = m.height; o.width = m.width; o.is_bigendian = m.is_bigendian; o.point_step = m.point_step;
  o.row_step = m.row_step; o.data = m.data; o.is_dense = m.is_dense;
  for (const auto& f : m.fields) { wire::PointField w; w.name = f.name; w.offset = f.offset; w.datatype = f.datatype; w.count = f.count; o.fields.push_back(w); }
  return o;
}
inline sensor_msgs::PointCloud2 to_ros(const wire::PointCloud2& m) {
  sensor_msgs::PointCloud2 o;
  o.header = to_ros(m.header); o.height = m.height; o.width = m.width; o.is_bigendian = m.is_bigendian; o.point_step = m.point_step;
  o.row_step = m.row_step; o.data = m.data; o.is_dense = m.is_dense;
  for (const auto& f : m.fields) { sensor_msgs::PointField w; w.name = f.name; w.offset = f.offset; w.datatype = f.datatype; w.count = f.count; o.fields.push_back(w); }
  return o;
}

inline wire::CloudInfoStamp from_ros(const ::rolo::CloudInfoStamp& m) {
  wire::CloudInfoStamp o;
  o.header = from_ros(m.header);
  o.startRingIndex.assign(m.startRingIndex.begin(), m.startRingIndex.end()); o.endRingIndex.assign(m.endRingIndex.begin(), m.endRingIndex.end());
  o.pointColInd.assign(m.pointColInd.begin(), m.pointColInd.end()); o.pointRange.assign(m.pointRange.begin(), m.pointRange.end());
  o.startOrientation = m.startOrientation; o.endOrientation = m.endOrientation; o.orientationDiff = m.orientationDiff;
  o.initialGuessX = m.initialGuessX; o.initialGuessY = m.initialGuessY; o.initialGuessZ = m.initialGuessZ;
  o.initialGuessRoll = m.initialGuessRoll; o.initialGuessPitch = m.initialGuessPitch; o.initialGuessYaw = m.initialGuessYaw;
  o.covariance.assign(m.covariance.begin(), m.covariance.end()); o.odomAvailable = m.odomAvailable;
  o.cloud_projected = from_ros(m.cloud_projected); o.extracted_corner = from_ros(m.extracted_corner); o.extracted_surface = from_ros(m.extracted_surface);
  o.extracted_normal = from_ros(m.extracted_normal); o.extracted_ground = from_ros(m.extracted_ground);
  return o;
}
inline ::rolo::CloudInfoStamp to_ros(const wire::CloudInfoStamp& m) {
  ::rolo::CloudInfoStamp o;
  o.header = to_ros(m.header);
  o.startRingIndex.assign(m.startRingIndex.begin(), m.startRingIndex.end()); o.endRingIndex.assign(m.endRingIndex.begin(), m.endRingIndex.end());
  o.pointColInd.assign(m.pointColInd.begin(), m.pointColInd.end()); o.pointRange.assign(m.pointRange.begin(), m.pointRange.end());
  o.startOrientation = m.startOrientation; o.endOrientation = m.endOrientation; o.orientationDiff = m.orientationDiff;
  o.initialGuessX = m.initialGuessX; o.initialGuessY = m.initialGuessY; o.initialGuessZ = m.initialGuessZ;
  o.initialGuessRoll = m.initialGuessRoll; o.initialGuessPitch = m.initialGuessPitch; o.initialGuessYaw = m.initialGuessYaw;
  o.covariance.assign(m.covariance.begin(), m.covariance.end()); o.odomAvailable = m.odomAvailable;
  o.cloud_projected = to_ros(m.cloud_projected); o.extracted_corner = to_ros(m.extracted_corner); o.extracted_surface = to_ros(m.extracted_surface);
  o.extracted_normal = to_ros(m.extracted_normal); o.extracted_ground = to_ros(m.extracted_ground);
  return o;
}

inline wire::Odometry from_ros(const nav_msgs::Odometry& m) {
  wire::Odometry o;
  o.header = from_ros(m.header); o.child_frame_id = m.child_frame_id;
  o.pose.position[0] = m.pose.pose.position.x; o.pose.position[1] = m.pose.pose.position.y; o.pose.position[2] = m.pose.pose.position.z;
  o.pose.orientation[0] = m.pose.pose.orientation.x; o.pose.orientation[1] = m.pose.pose.orientation.y; o.pose.orientation[2] = m.pose.pose.orientation.z;
  o.pose.orientation[3] = m.pose.pose.orientation.w;
  for (int i = 0; i < 36; i++) { o.pose_covariance[i] = m.pose.covariance[i]; o.twist_covariance[i] = m.twist.covariance[i]; }
  o.twist_linear[0] = m.twist.twist.linear.x; o.twist_linear[1] = m.twist.twist.linear.y; o.twist_linear[2] = m.twist.twist.linear.z;
  o.twist_angular[0] = m.twist.twist.angular.x; o.twist_angular[1] = m.twist.twist.angular.y; o.twist_angular[2] = m.twist.twist.angular.z;
  return o;
}
inline geometry_msgs::Pose to_ros(const wire::Pose& p) {
  geometry_msgs::Pose o;
  o.position.x = p.position[0]; o.position.y = p.position[1]; o.position.z = p.position[2];
  o.orientation.x = p.orientation[0]; o.orientation.y = p.orientation[1]; o.orientation.z = p.orientation[2]; o.orientation.w = p.orientation[3];
  return o;
}
inline nav_msgs::Odometry to_ros(const wire::Odometry& m) {
  nav_msgs::Odometry o;
  o.header = to_ros(m.header); o.child_frame_id = m.child_frame_id; o.pose.pose = to_ros(m.pose);
  for (int i = 0; i < 36; i++) { o.pose.covariance[i] = m.pose_covariance[i]; o.twist.covariance[i] = m.twist_covariance[i]; }
  o.twist.twist.linear.x = m.twist_linear[0]; o.twist.twist.linear.y = m.twist_linear[1]; o.twist.twist.linear.z = m.twist_linear[2];
  o.twist.twist.angular.x = m.twist_angular[0]; o.twist.twist.angular.y = m.twist_angular[1]; o.twist.twist.angular.z = m.twist_angular[2];
  return o;
}
inline geometry_msgs::PoseStamped to_ros(const wire::PoseStamped& m) { geometry_msgs::PoseStamped o; o.header = to_ros(m.header); o.pose = to_ros(m.pose); return o; }

inline nav_msgs::Path to_ros(const wire::Path& m) {
  nav_msgs::Path o;
  o.header = to_ros(m.header);
  o.poses.reserve(m.poses.size());
  for (const auto& p : m.poses) o.poses.push_back(to_ros(p));
  return o;
}
inline geometry_msgs::PoseWithCovarianceStamped to_ros(const wire::PoseWithCovarianceStamped& m) {
  geometry_msgs::PoseWithCovarianceStamped o;
  o.header = to_ros(m.header); o.pose.pose = to_ros(m.pose);
  for (int i = 0; i < 36; i++) o.pose.covariance[i] = m.covariance[i];
  return o;
}

// ParamLoader (include/rolo/utility.h:267-333): the keys these three nodes read, same names and defaults
inline NodeParams load_params(ros::NodeHandle& nh, bool& ok) {
  NodeParams P; ok = true;
  nh.param<std::string>("rolo/pointCloudTopic", P.pointCloudTopic, "points_raw");
  nh.param<std::string>("rolo/odomTopic", P.odomTopic, "odometry/imu");
  nh.param<std::string>("rolo/lidarFrame", P.lidarFrame, "base_link");
  nh.param<std::string>("rolo/baselinkFrame", P.baselinkFrame, "base_link");
  nh.param<std::string>("rolo/odometryFrame", P.odometryFrame, "odom");
  nh.param<std::string>("rolo/mapFrame", P.mapFrame, "map");
  std::string sensorStr;
  nh.param<std::string>("rolo/sensor", sensorStr, "");
  if (sensorStr == "velodyne") P.sensor = LidarType::VELODYNE;
  else if (sensorStr == "ouster") P.sensor = LidarType::OUSTER;
  else { ROS_ERROR_STREAM("Invalid sensor type (must be either 'velodyne' or 'ouster' or 'livox'): " << sensorStr); ok = false; }
  nh.param<int>("rolo/N_SCAN", P.N_SCAN, 16);
  nh.param<int>("rolo/Horizon_SCAN", P.Horizon_SCAN, 1800);
  nh.param<int>("rolo/downsampleRate", P.downsampleRate, 1);
  nh.param<float>("rolo/lidarMinRange", P.lidarMinRange, 1.0);
  nh.param<float>("rolo/lidarMaxRange", P.lidarMaxRange, 1000.0);
  nh.param<bool>("rolo/deskewEnabled", P.deskewEnabled, true);
  nh.param<float>("rolo/edgeThreshold", P.edgeThreshold, 0.1);
  nh.param<float>("rolo/surfThreshold", P.surfThreshold, 0.1);
  nh.param<float>("rolo/odometrySurfLeafSize", P.odometrySurfLeafSize, 0.2);
  nh.param<float>("rolo/continuousTrajectoryWeight", P.CT_lambda, 1.0);
  return P;
}

}  // namespace ros1
}  // namespace rolo
