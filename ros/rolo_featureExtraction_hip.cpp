// rolo_featureExtraction on MI355X — replaces src/featureExtraction.cpp of sdwyc/ROLO: same topics and queue sizes (:42-49, :290-301);
// the work is rolo::ros1::FeatureExtractionNode. Built only inside a catkin workspace; here it is type-checked against mock ROS headers (tests/test_ros_sources_compile.py).
#include "rolo_ros_convert.hpp"

class FeatureExtractionRos {
public:
  FeatureExtractionRos(ros::NodeHandle& nh, const rolo::ros1::NodeParams& P) : ctx_(0), node_(ctx_, P) {
    subLaserCloudInfo = nh.subscribe<rolo::CloudInfoStamp>("rolo/cloud_info", 1, &FeatureExtractionRos::laserCloudInfoHandler, this, ros::TransportHints().tcpNoDelay());
    pubLaserCloudInfo = nh.advertise<rolo::CloudInfoStamp>("rolo/feature/cloud_info", 1);
    pubCornerPoints = nh.advertise<sensor_msgs::PointCloud2>("rolo/feature/cloud_corner", 1);
    pubSurfacePoints = nh.advertise<sensor_msgs::PointCloud2>("rolo/feature/cloud_surface", 1);
    pubNormalPoints = nh.advertise<sensor_msgs::PointCloud2>("rolo/feature/cloud_normal", 1);
  }
  void laserCloudInfoHandler(const rolo::CloudInfoStampConstPtr& cloudIn) {
    rolo::wire::CloudInfoStamp out;
    try {
      if (node_.laserCloudInfoHandler(rolo::ros1::from_ros(*cloudIn), out) != rolo::ros1::Status::Published) { ROS_ERROR("rolo/cloud_info with inconsistent arrays"); return; }
    } catch (const rolo::Error& e) { ROS_ERROR_STREAM("rolo_featureExtraction (HIP): " << e.what()); return; }
    const rolo::CloudInfoStamp msg = rolo::ros1::to_ros(out);
    // publishCloud() publishes the viz clouds only when somebody listens (include/rolo/utility.h:441-442)
    if (pubCornerPoints.getNumSubscribers() != 0) pubCornerPoints.publish(msg.extracted_corner);
    if (pubSurfacePoints.getNumSubscribers() != 0) pubSurfacePoints.publish(msg.extracted_surface);
    if (pubNormalPoints.getNumSubscribers() != 0) pubNormalPoints.publish(msg.extracted_normal);
    pubLaserCloudInfo.publish(msg);
  }
private:
  rolo::Context ctx_;
  rolo::ros1::FeatureExtractionNode node_;
  ros::Subscriber subLaserCloudInfo;
  ros::Publisher pubLaserCloudInfo, pubCornerPoints, pubSurfacePoints, pubNormalPoints;
};

int main(int argc, char** argv) {
  ros::init(argc, argv, "rolo");
  ros::NodeHandle nh;
  bool ok = true;
  const rolo::ros1::NodeParams P = rolo::ros1::load_params(nh, ok);
  if (!ok) { ros::shutdown(); return 1; }
  FeatureExtractionRos FE(nh, P);
  ROS_INFO("\033[1;32m----> Feature Extraction Started (HIP).\033[0m");
  ros::spin();
  return 0;
}
