// Type check of the ROLO_HIP_WITH_PCL branch of include/rot_vgicp_hip.hpp against tests/cpp/mock_pcl: the explicit instantiation
// instantiates every member function of the drop-in class with pcl::PointCloud / Eigen types, as a ROLO catkin workspace would.
#define ROLO_HIP_WITH_PCL
#include "rot_vgicp_hip.hpp"

template class fast_gicp::RotVGICP<pcl::PointXYZI, pcl::PointXYZI>;

int main() {
  // never run on a machine without a GPU: constructing the class creates a device context
  return sizeof(fast_gicp::RotVGICP<pcl::PointXYZI, pcl::PointXYZI>) > 0 ? 0 : 1;
}
