// TransformFusion (src/lidarOdometry.cpp:47-323) as a roscpp class: subscribers / publishers / timers of :96-105, handlers :109-322 on
// rolo::ros1::TransformFusionNode (include/rolo_ros_nodes.hpp; rolo_fusion_* / PoseESEKF in librolo_hip.so). Shared by the two node
// sources that own one: rolo_lidarOdometry_hip.cpp (the reference's process layout) and rolo_front_fused_hip.cpp.
#pragma once
#include <mutex>

#include <autoware_rviz_msgs/Path.h>
#include <geometry_msgs/PoseWithCovarianceStamped.h>
#include <nav_msgs/Path.h>
#include <std_msgs/Float32.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_datatypes.h>
#include <tf/transform_listener.h>

#include "rolo_ros_convert.hpp"

class TransformFusionRos {
public:
  TransformFusionRos(ros::NodeHandle& nh, const rolo::ros1::NodeParams& P) : node_(P), P_(P) {
    if (P.lidarFrame != P.baselinkFrame) {   // :84-95
      try {
        tfListener.waitForTransform(P.lidarFrame, P.baselinkFrame, ros::Time(0), ros::Duration(3.0));
        tfListener.lookupTransform(P.lidarFrame, P.baselinkFrame, ros::Time(0), lidar2Baselink);
      } catch (tf::TransformException& ex) { ROS_ERROR("%s", ex.what()); }
    }
    subMappingOdometry = nh.subscribe<nav_msgs::Odometry>("rolo/mapping/odometry", 5, &TransformFusionRos::mappingOdometryHandler, this, ros::TransportHints().tcpNoDelay());
    subLidarOdometry = nh.subscribe<nav_msgs::Odometry>(P.odomTopic + "_incremental", 2000, &TransformFusionRos::lidarOdometryHandler, this, ros::TransportHints().tcpNoDelay());
    pubLidarOdometry = nh.advertise<nav_msgs::Odometry>(P.odomTopic, 2000);
    pubLidarPath = nh.advertise<nav_msgs::Path>("rolo/lidar_odometry/path", 1);
    pubLidarSpeed = nh.advertise<std_msgs::Float32>(P.odomTopic + "/speed", 2000);
    pubFuturePath = nh.advertise<autoware_rviz_msgs::Path>("future_path", 1);
    pubFuturePoseLidar = nh.advertise<geometry_msgs::PoseWithCovarianceStamped>("future_pose_lidar", 1);
    fusionTimer = nh.createTimer(ros::Duration(1.0 / 20.0), &TransformFusionRos::fusionTimerHandler, this);
    predictTimer = nh.createTimer(ros::Duration(1.0 / 30.0), &TransformFusionRos::predictTimerHandler, this);
  }
  void mappingOdometryHandler(const nav_msgs::OdometryConstPtr& odomMsg) {
    std::lock_guard<std::mutex> lock(mtx);
    node_.mappingOdometryHandler(rolo::ros1::from_ros(*odomMsg));
  }
  void lidarOdometryHandler(const nav_msgs::OdometryConstPtr& odomMsg) {
    std::lock_guard<std::mutex> lock(mtx);
    node_.lidarOdometryHandler(rolo::ros1::from_ros(*odomMsg));
  }
  void fusionTimerHandler(const ros::TimerEvent&) {
    std::lock_guard<std::mutex> lock(mtx);
    const ros::Time stamp = ros::Time::now();
    tfMap2Odom.sendTransform(tf::StampedTransform(tf::Transform(tf::createQuaternionFromRPY(0, 0, 0), tf::Vector3(0, 0, 0)), stamp, P_.mapFrame, P_.odometryFrame));
    rolo::wire::Time now; now.sec = stamp.sec; now.nsec = stamp.nsec;
    rolo::ros1::TransformFusionNode::FusionOutputs o;
    if (!node_.fusionTimerHandler(now, o)) return;
    const nav_msgs::Odometry laserOdometry = rolo::ros1::to_ros(o.odometry);
    pubLidarOdometry.publish(laserOdometry);
    tf::Transform tCur;   // odom -> base_link (:212-219)
    tf::poseMsgToTF(laserOdometry.pose.pose, tCur);
    if (P_.lidarFrame != P_.baselinkFrame) tCur = tCur * lidar2Baselink;
    tfOdom2BaseLink.sendTransform(tf::StampedTransform(tCur, stamp, P_.odometryFrame, P_.baselinkFrame));
    if (o.path_updated && pubLidarPath.getNumSubscribers() != 0) pubLidarPath.publish(rolo::ros1::to_ros(node_.path));
    std_msgs::Float32 speed_msg;
    speed_msg.data = o.speed.data;
    pubLidarSpeed.publish(speed_msg);
  }
  void predictTimerHandler(const ros::TimerEvent&) {
    std::lock_guard<std::mutex> lock(mtx);
    const ros::Time stamp = ros::Time::now();
    rolo::wire::Time now; now.sec = stamp.sec; now.nsec = stamp.nsec;
    rolo::ros1::TransformFusionNode::PredictOutputs o;
    if (!node_.predictTimerHandler(now, o)) return;
    autoware_rviz_msgs::Path future_path;
    future_path.header = rolo::ros1::to_ros(o.header);
    future_path.points.reserve(o.points.size());
    for (const rolo_future_point& p : o.points) {
      autoware_rviz_msgs::PathPoint pt;
      pt.pose.position.x = p.position[0]; pt.pose.position.y = p.position[1]; pt.pose.position.z = p.position[2];
      pt.pose.orientation.x = p.orientation[0]; pt.pose.orientation.y = p.orientation[1]; pt.pose.orientation.z = p.orientation[2]; pt.pose.orientation.w = p.orientation[3];
      pt.longitudinal_velocity_mps = p.longitudinal_velocity_mps; pt.lateral_velocity_mps = p.lateral_velocity_mps; pt.heading_rate_rps = p.heading_rate_rps;
      pt.is_final = p.is_final != 0;
      future_path.points.push_back(pt);
    }
    pubFuturePath.publish(future_path);
    pubFuturePoseLidar.publish(rolo::ros1::to_ros(o.future_pose_lidar));
  }
private:
  rolo::ros1::TransformFusionNode node_;
  rolo::ros1::NodeParams P_;
  std::mutex mtx;
  tf::TransformListener tfListener;
  tf::StampedTransform lidar2Baselink;
  tf::TransformBroadcaster tfMap2Odom, tfOdom2BaseLink;
  ros::Subscriber subLidarOdometry, subMappingOdometry;
  ros::Publisher pubLidarOdometry, pubLidarPath, pubLidarSpeed, pubFuturePath, pubFuturePoseLidar;
  ros::Timer fusionTimer, predictTimer;
};

