#pragma once
namespace pcl { struct PointXYZI { float x, y, z, data3, intensity, pad[3]; }; static_assert(sizeof(PointXYZI) == 32, "pcl::PointXYZI is 32 bytes"); }
