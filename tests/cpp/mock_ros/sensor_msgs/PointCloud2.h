#pragma once
#include <std_msgs/Header.h>
#include <vector>
namespace sensor_msgs {
struct PointField { std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0; };
struct PointCloud2 {
  std_msgs::Header header; uint32_t height = 0, width = 0; std::vector<PointField> fields; uint8_t is_bigendian = 0;
  uint32_t point_step = 0, row_step = 0; std::vector<uint8_t> data; uint8_t is_dense = 0;
};
typedef boost::shared_ptr<PointCloud2 const> PointCloud2ConstPtr;
}  // namespace sensor_msgs
