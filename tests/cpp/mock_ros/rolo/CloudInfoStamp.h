// the generated class of msg/CloudInfoStamp.msg (field order and types of the message definition)
#pragma once
#include <sensor_msgs/PointCloud2.h>
namespace rolo {
struct CloudInfoStamp {
  std_msgs::Header header;
  std::vector<int32_t> startRingIndex, endRingIndex, pointColInd;
  std::vector<float> pointRange;
  float startOrientation = 0, endOrientation = 0, orientationDiff = 0;
  float initialGuessX = 0, initialGuessY = 0, initialGuessZ = 0, initialGuessRoll = 0, initialGuessPitch = 0, initialGuessYaw = 0;
  std::vector<float> covariance;
  uint8_t odomAvailable = 0;
  sensor_msgs::PointCloud2 cloud_projected, extracted_corner, extracted_surface, extracted_normal, extracted_ground;
};
typedef boost::shared_ptr<CloudInfoStamp const> CloudInfoStampConstPtr;
}  // namespace rolo
