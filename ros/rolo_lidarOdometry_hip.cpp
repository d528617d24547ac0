// rolo_lidarOdometry on MI355X — replaces src/lidarOdometry.cpp of sdwyc/ROLO, BOTH classes its main() instantiates (:715-729):
//   LidarOdometryRos    = LidarOdometry (:325-713): same topics, queue sizes, frames and TF (:394-405, :645-697); the work is
//                         rolo::ros1::LidarOdometryNode on the HIP registration;
//   TransformFusionRos  = TransformFusion (:47-323): subscribers / publishers / timers of :96-105, handlers :109-322 on
//                         rolo::ros1::TransformFusionNode (rolo_fusion_* / PoseESEKF in librolo_hip.so) — the 20 Hz ESKF-smoothed odomTopic every
//                         downstream consumer reads, odomTopic + "/speed", the 1 s path, future_path / future_pose_lidar at 30 Hz, the
//                         map -> odom and odom -> base_link transforms.
// Built only inside a catkin workspace; here it is type-checked against mock ROS headers (tests/test_ros_sources_compile.py).
#include <algorithm>
#include <mutex>

#include <nav_msgs/Path.h>
#include <std_msgs/Float64MultiArray.h>
#include <tf/transform_broadcaster.h>

#include "rolo_ros_convert.hpp"
#include "rolo_transform_fusion_ros.hpp"

class LidarOdometryRos {
public:
  LidarOdometryRos(ros::NodeHandle& nh, const rolo::ros1::NodeParams& P) : ctx_(0), node_(ctx_, P), P_(P) {
    subOdometryMapped = nh.subscribe<nav_msgs::Odometry>("rolo/mapping/odometry", 10, &LidarOdometryRos::odometryHandler, this, ros::TransportHints().tcpNoDelay());
    subCloudInfo = nh.subscribe<rolo::CloudInfoStamp>("rolo/feature/cloud_info", 10, &LidarOdometryRos::cloudHandler, this, ros::TransportHints().tcpNoDelay());
    pubFrontCloudInfo = nh.advertise<rolo::CloudInfoStamp>(P.odomTopic + "/cloud_info", 2000);
    pubLidarOdometry = nh.advertise<nav_msgs::Odometry>(P.odomTopic + "_incremental", 2000);
    pubLidarPose = nh.advertise<geometry_msgs::PoseStamped>(P.odomTopic + "_incremental/pose", 2000);
    pubLaserPath = nh.advertise<nav_msgs::Path>(P.odomTopic + "_incremental/path", 2000);
    pubRegScan = nh.advertise<sensor_msgs::PointCloud2>(P.odomTopic + "/registration_scan", 10);
    pubPlotData = nh.advertise<std_msgs::Float64MultiArray>("rolo/data_test", 10);   // :405: advertised, never published by the reference either — part of the node's graph
  }
  void odometryHandler(const nav_msgs::OdometryConstPtr& mappedOdom) {
    std::lock_guard<std::mutex> lock(mtx);
    node_.odometryHandler(rolo::ros1::from_ros(*mappedOdom));
  }
  void cloudHandler(const rolo::CloudInfoStampConstPtr& cloudIn) {
    rolo::ros1::LidarOdometryNode::Outputs o;
    rolo::ros1::Status st;
    const ros::Time now = ros::Time::now();
    rolo::wire::Time wnow; wnow.sec = now.sec; wnow.nsec = now.nsec;
    try {
      std::lock_guard<std::mutex> lock(mtx);
      st = node_.cloudHandler(rolo::ros1::from_ros(*cloudIn), wnow, o);
    } catch (const rolo::Error& e) { ROS_ERROR_STREAM("rolo_lidarOdometry (HIP): " << e.what()); return; }
    if (st != rolo::ros1::Status::Published) return;   // first frame: stash only
    // pubMessage :655-697
    if (pubRegScan.getNumSubscribers() != 0) pubRegScan.publish(rolo::ros1::to_ros(o.registration_scan));
    const geometry_msgs::PoseStamped laser_pose = rolo::ros1::to_ros(o.laser_pose);
    pubLidarPose.publish(laser_pose);
    laser_odom_path.header = laser_pose.header;
    laser_odom_path.poses.push_back(laser_pose);
    nav_msgs::Path reversed = laser_odom_path;
    std::reverse(reversed.poses.begin(), reversed.poses.end());
    pubLaserPath.publish(reversed);
    pubLidarOdometry.publish(rolo::ros1::to_ros(o.laser_odom_incremental));
    pubFrontCloudInfo.publish(rolo::ros1::to_ros(o.odometry_cloud));
    // pubTranform :645-653 — only after a registered frame (:560-563)
    if (o.frame == rolo::LidarOdometry::Registered) {
      static tf::TransformBroadcaster br;
      const auto& p = o.laser_pose.pose;
      tf::Transform t(tf::Quaternion(p.orientation[0], p.orientation[1], p.orientation[2], p.orientation[3]), tf::Vector3(p.position[0], p.position[1], p.position[2]));
      br.sendTransform(tf::StampedTransform(t, laser_pose.header.stamp, P_.odometryFrame, "lidar"));
    }
  }
private:
  rolo::Context ctx_;
  rolo::ros1::LidarOdometryNode node_;
  rolo::ros1::NodeParams P_;
  std::mutex mtx;
  nav_msgs::Path laser_odom_path;
  ros::Subscriber subOdometryMapped, subCloudInfo;
  ros::Publisher pubFrontCloudInfo, pubLidarOdometry, pubLidarPose, pubLaserPath, pubRegScan, pubPlotData;
};

int main(int argc, char** argv) {
  ros::init(argc, argv, "rolo");
  ros::NodeHandle nh;
  bool ok = true;
  const rolo::ros1::NodeParams P = rolo::ros1::load_params(nh, ok);
  if (!ok) { ros::shutdown(); return 1; }
  ROS_INFO("\033[1;32m----> Laser Odometry Started (HIP).\033[0m");
  LidarOdometryRos LO(nh, P);
  TransformFusionRos TF(nh, P);   // lidarOdometry.cpp:720-721: both objects in one process, two spinner threads
  ros::MultiThreadedSpinner spinner(2);
  spinner.spin();
  return 0;
}
