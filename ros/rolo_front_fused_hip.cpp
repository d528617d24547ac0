// rolo_front_fused on MI355X — SURVEY.md 8f.2: the three front-end processes of launch/module_loam.launch:6-9 (rolo_imageProjection,
// rolo_featureExtraction, rolo_lidarOdometry) as ONE process. Same input and output topics as the chain — pointCloudTopic and
// rolo/mapping/odometry in; odomTopic + "_incremental" (+ "/pose", "/path"), odomTopic + "/cloud_info", TF odometryFrame -> "lidar" and,
// through TransformFusionRos, odomTopic, odomTopic + "/speed", rolo/lidar_odometry/path, future_path, future_pose_lidar out — but the two
// rolo/CloudInfoStamp hops between the nodes (rolo/cloud_info, rolo/feature/cloud_info: a serialisation, a TCP copy and a host <-> device copy
// of every cloud each) are gone: the PointCloud2 payload is unpacked on the device and the range image, the feature clouds and both
// registration inputs stay in HBM (rolo::ros1::FusedFrontEndNode on rolo_odom_submit_msg / rolo_odom_collect). Launch it INSTEAD of the
// three nodes; the intermediate topics are then not published (nothing outside the three nodes subscribes to them in the reference's launch
// files). Built only inside a catkin workspace; type-checked here against mock ROS headers (tests/test_ros_sources_compile.py).
#include <algorithm>
#include <mutex>

#include <nav_msgs/Path.h>
#include <tf/transform_broadcaster.h>

#include "rolo_ros_convert.hpp"
#include "rolo_transform_fusion_ros.hpp"

class FusedFrontEndRos {
public:
  FusedFrontEndRos(ros::NodeHandle& nh, const rolo::ros1::NodeParams& P) : ctx_(0), node_(ctx_, P), P_(P) {
    subLaserCloud = nh.subscribe<sensor_msgs::PointCloud2>(P.pointCloudTopic, 10, &FusedFrontEndRos::cloudHandler, this, ros::TransportHints().tcpNoDelay());   // imageProjection.cpp:89
    subOdometryMapped = nh.subscribe<nav_msgs::Odometry>("rolo/mapping/odometry", 10, &FusedFrontEndRos::odometryHandler, this, ros::TransportHints().tcpNoDelay());   // lidarOdometry.cpp:396
    pubFrontCloudInfo = nh.advertise<rolo::CloudInfoStamp>(P.odomTopic + "/cloud_info", 2000);
    pubLidarOdometry = nh.advertise<nav_msgs::Odometry>(P.odomTopic + "_incremental", 2000);
    pubLidarPose = nh.advertise<geometry_msgs::PoseStamped>(P.odomTopic + "_incremental/pose", 2000);
    pubLaserPath = nh.advertise<nav_msgs::Path>(P.odomTopic + "_incremental/path", 2000);
  }
  void odometryHandler(const nav_msgs::OdometryConstPtr& mappedOdom) {
    std::lock_guard<std::mutex> lock(mtx);
    node_.odometryHandler(rolo::ros1::from_ros(*mappedOdom));
  }
  void cloudHandler(const sensor_msgs::PointCloud2ConstPtr& laserCloudMsg) {
    rolo::ros1::FusedFrontEndNode::Outputs o;
    rolo::ros1::Status st;
    try {
      std::lock_guard<std::mutex> lock(mtx);
      st = node_.cloudHandler(rolo::ros1::from_ros(*laserCloudMsg), o);
    } catch (const rolo::Error& e) { ROS_ERROR_STREAM("rolo_front_fused (HIP): " << e.what()); return; }
    if (st == rolo::ros1::Status::NonDense) { ROS_ERROR("Point cloud is not in dense format, please remove NaN points first!"); ros::shutdown(); return; }   // imageProjection.cpp:226-231
    if (st == rolo::ros1::Status::BadFields || st == rolo::ros1::Status::BadSensor) { ROS_ERROR("rolo_front_fused (HIP): unusable point cloud (fields / sensor type)"); ros::shutdown(); return; }
    if (st != rolo::ros1::Status::Published) return;
    // pubMessage, lidarOdometry.cpp:655-697
    const geometry_msgs::PoseStamped laser_pose = rolo::ros1::to_ros(o.laser_pose);
    pubLidarPose.publish(laser_pose);
    laser_odom_path.header = laser_pose.header;
    laser_odom_path.poses.push_back(laser_pose);
    nav_msgs::Path reversed = laser_odom_path;
    std::reverse(reversed.poses.begin(), reversed.poses.end());
    pubLaserPath.publish(reversed);
    pubLidarOdometry.publish(rolo::ros1::to_ros(o.laser_odom_incremental));
    pubFrontCloudInfo.publish(rolo::ros1::to_ros(o.odometry_cloud));
    if (o.frame == rolo::LidarOdometry::Registered) {   // pubTranform :645-653
      static tf::TransformBroadcaster br;
      const auto& p = o.laser_pose.pose;
      tf::Transform t(tf::Quaternion(p.orientation[0], p.orientation[1], p.orientation[2], p.orientation[3]), tf::Vector3(p.position[0], p.position[1], p.position[2]));
      br.sendTransform(tf::StampedTransform(t, laser_pose.header.stamp, P_.odometryFrame, "lidar"));
    }
  }
private:
  rolo::Context ctx_;
  rolo::ros1::FusedFrontEndNode node_;
  rolo::ros1::NodeParams P_;
  std::mutex mtx;
  nav_msgs::Path laser_odom_path;
  ros::Subscriber subLaserCloud, subOdometryMapped;
  ros::Publisher pubFrontCloudInfo, pubLidarOdometry, pubLidarPose, pubLaserPath;
};

int main(int argc, char** argv) {
  ros::init(argc, argv, "rolo");
  ros::NodeHandle nh;
  bool ok = true;
  const rolo::ros1::NodeParams P = rolo::ros1::load_params(nh, ok);
  if (!ok) { ros::shutdown(); return 1; }
  ROS_INFO("\033[1;32m----> Fused front end Started (HIP): image projection + feature extraction + laser odometry in one process.\033[0m");
  FusedFrontEndRos FE(nh, P);
  TransformFusionRos TF(nh, P);
  ros::MultiThreadedSpinner spinner(3);
  spinner.spin();
  return 0;
}
