// rolo_imageProjection on MI355X — replaces src/imageProjection.cpp of sdwyc/ROLO: same node name, topics, queue sizes and transport
// hints (:86-93, :515-528); everything between "message arrived" and "publish" is rolo::ros1::ImageProjectionNode
// (include/rolo_ros_nodes.hpp -> librolo_hip.so). Built only inside a catkin workspace (see ros/README.md); here it is type-checked against mock ROS headers (tests/test_ros_sources_compile.py).
#include <mutex>

#include "rolo_ros_convert.hpp"

class ImageProjectionRos {
public:
  ImageProjectionRos(ros::NodeHandle& nh, const rolo::ros1::NodeParams& P) : ctx_(0), node_(ctx_, P) {
    subLaserCloud = nh.subscribe<sensor_msgs::PointCloud2>(P.pointCloudTopic, 10, &ImageProjectionRos::cloudHandler, this, ros::TransportHints().tcpNoDelay());
    subOdom = nh.subscribe<nav_msgs::Odometry>(P.odomTopic + "_incremental", 2000, &ImageProjectionRos::odometryHandler, this, ros::TransportHints().tcpNoDelay());
    pubLaserCloudInfo = nh.advertise<rolo::CloudInfoStamp>("rolo/cloud_info", 1);
  }
  void odometryHandler(const nav_msgs::OdometryConstPtr& odomMsg) {
    std::lock_guard<std::mutex> lock(odomLock);
    node_.odometryHandler(rolo::ros1::from_ros(*odomMsg));
  }
  void cloudHandler(const sensor_msgs::PointCloud2ConstPtr& laserCloudMsg) {
    rolo::wire::CloudInfoStamp out;
    rolo::ros1::Status st;
    try {
      std::lock_guard<std::mutex> lock(odomLock);   // the node core reads the odometry queue (3 spinner threads)
      st = node_.cloudHandler(rolo::ros1::from_ros(*laserCloudMsg), out);
    } catch (const rolo::Error& e) { ROS_ERROR_STREAM("rolo_imageProjection (HIP): " << e.what()); return; }
    if (st == rolo::ros1::Status::NonDense) { ROS_ERROR("Point cloud is not in dense format, please remove NaN points first!"); ros::shutdown(); return; }
    if (st == rolo::ros1::Status::BadSensor) { ROS_ERROR_STREAM("Unknown sensor type: " << int(node_.P.sensor)); ros::shutdown(); return; }
    if (st == rolo::ros1::Status::BadFields) { ROS_ERROR("Point cloud without x / y / z / ring fields"); ros::shutdown(); return; }
    if (st == rolo::ros1::Status::Published) pubLaserCloudInfo.publish(rolo::ros1::to_ros(out));
  }
private:
  rolo::Context ctx_;
  rolo::ros1::ImageProjectionNode node_;
  std::mutex odomLock;
  ros::Subscriber subLaserCloud, subOdom;
  ros::Publisher pubLaserCloudInfo;
};

int main(int argc, char** argv) {
  ros::init(argc, argv, "image_projection");
  ros::NodeHandle nh;
  bool ok = true;
  const rolo::ros1::NodeParams P = rolo::ros1::load_params(nh, ok);
  if (!ok) { ros::shutdown(); return 1; }
  ImageProjectionRos IP(nh, P);
  ROS_INFO("\033[1;32m----> Image Projection Started (HIP).\033[0m");
  ros::MultiThreadedSpinner spinner(3);
  spinner.spin();
  return 0;
}
