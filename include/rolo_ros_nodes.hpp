// rolo_ros_nodes.hpp — the message-level halves of the three front-end ROS nodes of sdwyc/ROLO, ROS-free: what each node does
// between "a message arrived" and "publish", on the wire structs of rolo_ros_wire.hpp and the HIP node cores of rolo_nodes_hip.hpp.
// The catkin node sources (ros/*.cpp) only convert between ROS message classes and these structs and own subscribers / publishers.
//
//   rolo::ros1::ImageProjectionNode    cloudHandler / odometryHandler          src/imageProjection.cpp:86-93, 150-176, 179-366, 507-512
//   rolo::ros1::FeatureExtractionNode  laserCloudInfoHandler                   src/featureExtraction.cpp:42-49, 71-85, 268-287
//   rolo::ros1::LidarOdometryNode      odometryHandler / cloudHandler / pubMessage   src/lidarOdometry.cpp:394-405, 440-446, 503-570, 655-697
//
// Where the reference calls ros::shutdown() (non-dense cloud, unknown sensor) the handlers return a Status and leave the policy to
// the node source. Deviation, stated: a cloud without a `ring` field is refused (the reference would silently put every point in row 0).
#pragma once
#include <cmath>
#include <deque>
#include <queue>
#include <string>
#include <vector>

#include "rolo_nodes_hip.hpp"
#include "rolo_ros_wire.hpp"

namespace rolo {
namespace ros1 {

enum class LidarType { VELODYNE = 0, OUSTER = 1 };

// the keys ParamLoader reads for these three nodes (include/rolo/utility.h:267-333), with its defaults
struct NodeParams {
  std::string pointCloudTopic = "points_raw", odomTopic = "odometry/imu";
  std::string lidarFrame = "base_link", baselinkFrame = "base_link", odometryFrame = "odom";
  LidarType sensor = LidarType::VELODYNE;
  int N_SCAN = 16, Horizon_SCAN = 1800, downsampleRate = 1;
  float lidarMinRange = 1.0f, lidarMaxRange = 1000.0f;
  bool deskewEnabled = true;
  float edgeThreshold = 0.1f, surfThreshold = 0.1f, odometrySurfLeafSize = 0.2f;
  float CT_lambda = 1.0f;
  FrontParams front() const { return FrontParams(N_SCAN, Horizon_SCAN, downsampleRate, lidarMinRange, lidarMaxRange, edgeThreshold, surfThreshold, odometrySurfLeafSize); }
};

enum class Status { Published = 0, Queued = 1 /* cachePointCloud: fewer than three clouds yet */, FirstFrame = 2, NonDense = -1, BadFields = -2, BadSensor = -3 };

// one scalar of a PointCloud2 record as double (pcl::moveFromROSMsg's field mapping converts nothing: types must match the struct; the
// node cores accept any numeric datatype for ring / time, which covers the Velodyne and the Ouster drivers)
inline double field_value(const uint8_t* rec, const wire::PointField& f) {
  const uint8_t* p = rec + f.offset;
  switch (f.datatype) {
    case wire::PointField::INT8: { int8_t v; std::memcpy(&v, p, 1); return v; }
    case wire::PointField::UINT8: { uint8_t v; std::memcpy(&v, p, 1); return v; }
    case wire::PointField::INT16: { int16_t v; std::memcpy(&v, p, 2); return v; }
    case wire::PointField::UINT16: { uint16_t v; std::memcpy(&v, p, 2); return v; }
    case wire::PointField::INT32: { int32_t v; std::memcpy(&v, p, 4); return v; }
    case wire::PointField::UINT32: { uint32_t v; std::memcpy(&v, p, 4); return v; }
    case wire::PointField::FLOAT32: { float v; std::memcpy(&v, p, 4); return v; }
    case wire::PointField::FLOAT64: { double v; std::memcpy(&v, p, 8); return v; }
  }
  return 0.0;
}

class ImageProjectionNode {
public:
  ImageProjectionNode(Context& ctx, const NodeParams& p) : P(p), fp_(p.front()), core_(ctx, fp_) {
    // allocateMemory(): the four index arrays are members sized once; entries behind N keep what earlier frames left there
    cloudInfoStamp.startRingIndex.assign(P.N_SCAN, 0); cloudInfoStamp.endRingIndex.assign(P.N_SCAN, 0);
    cloudInfoStamp.pointColInd.assign((size_t)P.N_SCAN * P.Horizon_SCAN, 0); cloudInfoStamp.pointRange.assign((size_t)P.N_SCAN * P.Horizon_SCAN, 0.f);
  }
  // odometryHandler :150-156 — odomTopic + "_incremental", feeds the de-skew
  void odometryHandler(const wire::Odometry& odomMsg) { odomQueue.push_back(odomMsg); if (odomQueue.size() >= 2) odomAvailable = true; }

  // cloudHandler :158-176. On Status::Published `out` is the rolo/cloud_info message.
  Status cloudHandler(const wire::PointCloud2& laserCloudMsg, wire::CloudInfoStamp& out) {
    // ---- cachePointCloud :179-263 ----
    cloudQueue.push_back(laserCloudMsg);
    if (cloudQueue.size() <= 2) return Status::Queued;
    currentCloudMsg = std::move(cloudQueue.front());
    cloudQueue.pop_front();
    if (P.sensor != LidarType::VELODYNE && P.sensor != LidarType::OUSTER) return Status::BadSensor;
    if (P.sensor == LidarType::OUSTER) timeField = "t";
    const wire::PointCloud2& m = currentCloudMsg;
    const wire::PointField *fx = m.field("x"), *fy = m.field("y"), *fz = m.field("z"), *fring = m.field("ring"), *ftime = m.field(timeField);
    const size_t n = m.size();
    if (!fx || !fy || !fz || !fring || fx->datatype != wire::PointField::FLOAT32 || fy->datatype != wire::PointField::FLOAT32 ||
        fz->datatype != wire::PointField::FLOAT32 || m.point_step == 0 || m.data.size() < n * (size_t)m.point_step)
      return Status::BadFields;
    xyz_.resize(n * 3); ring_.resize(n); time_.assign(n, 0.f);
    for (size_t i = 0; i < n; i++) {   // pcl::moveFromROSMsg (+ the Ouster conversion loop :196-207: time = t * 1e-9f)
      const uint8_t* rec = m.data.data() + i * (size_t)m.point_step;
      std::memcpy(&xyz_[3 * i], rec + fx->offset, 4); std::memcpy(&xyz_[3 * i + 1], rec + fy->offset, 4); std::memcpy(&xyz_[3 * i + 2], rec + fz->offset, 4);
      ring_[i] = (uint16_t)field_value(rec, *fring);
      if (ftime) time_[i] = P.sensor == LidarType::OUSTER ? (float)(uint32_t)field_value(rec, *ftime) * 1e-9f : (float)field_value(rec, *ftime);
    }
    cloudHeader = m.header;
    timeScanCur = cloudHeader.stamp.toSec();
    timeScanEnd = timeScanCur + (n ? (double)time_[n - 1] : 0.0);
    if (!m.is_dense) return Status::NonDense;   // "Point cloud is not in dense format, please remove NaN points first!" + ros::shutdown()
    if (ringFlag == 0) ringFlag = 1;             // a missing ring field was refused above
    if (timeFlag == 0) timeFlag = ftime ? 1 : -1;
    // ---- deskewCloudInfo :266-366 ----
    if (P.deskewEnabled && odomAvailable && n > 0) {
      const double gate = timeFlag == -1 ? 0.25 : 0.3;
      while (!odomQueue.empty()) { if (std::fabs(timeScanCur - odomQueue.front().header.stamp.toSec()) > gate) odomQueue.pop_front(); else break; }
      if (!odomQueue.empty()) {
        float front6[6], back6[6], incre6[6];
        odom2pose(odomQueue.front(), front6); odom2pose(odomQueue.back(), back6);
        rolo_odom_increment(front6, back6, incre6);
        odomTimeDiff = odomQueue.back().header.stamp.toSec() - odomQueue.front().header.stamp.toSec();
        if (odomTimeDiff != 0.0) {
          if (timeFlag == -1) core_.setDeskew(incre6 + 3, scanPeriod, odomTimeDiff, nullptr, 0);   // times interpolated from the azimuth (:270-327)
          else { rel_.resize(n); for (size_t i = 0; i < n; i++) rel_[i] = std::fabs(time_[i]); core_.setDeskew(incre6 + 3, scanPeriod, odomTimeDiff, rel_.data(), (int)n); }
        }
      }
    }
    // ---- projectPointCloud + cloudExtraction :399-505 (HIP) ----
    const CloudInfo& ci = core_.projectPointCloud(xyz_.data(), 3, ring_.data(), (int)n);
    std::copy(ci.startRingIndex.begin(), ci.startRingIndex.end(), cloudInfoStamp.startRingIndex.begin());
    std::copy(ci.endRingIndex.begin(), ci.endRingIndex.end(), cloudInfoStamp.endRingIndex.begin());
    std::copy(ci.pointColInd.begin(), ci.pointColInd.end(), cloudInfoStamp.pointColInd.begin());
    std::copy(ci.pointRange.begin(), ci.pointRange.end(), cloudInfoStamp.pointRange.begin());
    // ---- publishClouds :507-512 ----
    cloudInfoStamp.header = cloudHeader;
    cloudInfoStamp.cloud_projected = wire::toROSMsgXYZI(ci.extractedCloud.data(), (size_t)ci.n_valid, cloudHeader.stamp, P.lidarFrame);
    out = cloudInfoStamp;
    return Status::Published;
  }

  NodeParams P;
  wire::CloudInfoStamp cloudInfoStamp;
  wire::Header cloudHeader;
  double timeScanCur = 0, timeScanEnd = 0, odomTimeDiff = -1.0;
  float scanPeriod = 0.1f;
  int ringFlag = 0, timeFlag = 0;
  bool odomAvailable = false;
  std::string timeField = "time";
  std::deque<wire::PointCloud2> cloudQueue;
  std::deque<wire::Odometry> odomQueue;

private:
  // odom2affine :138-148 as x, y, z, roll, pitch, yaw (pcl::getTransformation takes floats)
  static void odom2pose(const wire::Odometry& o, float pose6[6]) {
    double r, p, y; wire::getRPY(o.pose.orientation, r, p, y);
    pose6[0] = (float)o.pose.position[0]; pose6[1] = (float)o.pose.position[1]; pose6[2] = (float)o.pose.position[2];
    pose6[3] = (float)r; pose6[4] = (float)p; pose6[5] = (float)y;
  }
  FrontParams fp_;
  ImageProjection core_;
  wire::PointCloud2 currentCloudMsg;
  std::vector<float> xyz_, time_, rel_;
  std::vector<uint16_t> ring_;
};

class FeatureExtractionNode {
public:
  FeatureExtractionNode(Context& ctx, const NodeParams& p) : P(p), ctx_(ctx), fp_(p.front()), core_(ctx, fp_) {}
  // laserCloudInfoHandler :71-85 + publishFeatureCloud :276-287; `out` is the rolo/feature/cloud_info message
  Status laserCloudInfoHandler(const wire::CloudInfoStamp& cloudIn, wire::CloudInfoStamp& out) {
    cloudInfo = cloudIn;
    cloudHeader = cloudIn.header;
    if (!wire::fromROSMsgXYZI(cloudIn.cloud_projected, extracted_)) return Status::BadFields;
    const int n = (int)(extracted_.size() / 4);
    if ((int)cloudIn.startRingIndex.size() < P.N_SCAN || (int)cloudIn.endRingIndex.size() < P.N_SCAN || (int)cloudIn.pointColInd.size() < n ||
        (int)cloudIn.pointRange.size() < n)
      return Status::BadFields;
    check(rolo_front_load_projection(ctx_.get(), &fp_, extracted_.data(), cloudIn.pointColInd.data(), cloudIn.pointRange.data(), cloudIn.startRingIndex.data(),
                                     cloudIn.endRingIndex.data(), n), "rolo_front_load_projection");
    core_.extractFeatures(n);
    // freeCloudInfoMemory :268-274
    cloudInfo.startRingIndex.clear(); cloudInfo.endRingIndex.clear(); cloudInfo.pointColInd.clear(); cloudInfo.pointRange.clear();
    cloudInfo.extracted_corner = wire::toROSMsgXYZI(core_.cornerCloud.data(), core_.cornerCloud.size() / 4, cloudHeader.stamp, P.lidarFrame);
    cloudInfo.extracted_surface = wire::toROSMsgXYZI(core_.surfaceCloud.data(), core_.surfaceCloud.size() / 4, cloudHeader.stamp, P.lidarFrame);
    cloudInfo.extracted_normal = wire::toROSMsgXYZI(nullptr, 0, cloudHeader.stamp, P.lidarFrame);   // normalCloud stays empty (:153-266 never fills it)
    out = cloudInfo;
    return Status::Published;
  }
  NodeParams P;
  wire::CloudInfoStamp cloudInfo;
  wire::Header cloudHeader;
private:
  Context& ctx_;
  FrontParams fp_;
  FeatureExtraction core_;
  std::vector<float> extracted_;
};

class LidarOdometryNode {
public:
  struct Outputs {   // what pubMessage publishes (:655-697)
    wire::Odometry laser_odom_incremental;      // odomTopic + "_incremental"
    wire::PoseStamped laser_pose;               // odomTopic + "_incremental/pose"
    wire::CloudInfoStamp odometry_cloud;        // odomTopic + "/cloud_info"
    wire::PointCloud2 registration_scan;        // odomTopic + "/registration_scan"
    LidarOdometry::Status frame = LidarOdometry::FirstFrame;
  };
  LidarOdometryNode(Context& ctx, const NodeParams& p) : P(p), ctx_(ctx), core_(ctx, p.CT_lambda) {}
  // odometryHandler :440-446 — rolo/mapping/odometry from the back end (gates scan matching)
  void odometryHandler(const wire::Odometry& mappedOdom) { core_.odometryHandler(mappedOdom.header.stamp.toSec()); }

  // cloudHandler :503-570; `now` = ros::Time::now(). Status::FirstFrame: nothing is published.
  Status cloudHandler(const wire::CloudInfoStamp& cloudIn, const wire::Time& now, Outputs& out) {
    cloudTimeStamp = cloudIn.header.stamp;
    laserCloudInfoBuf.push(cloudIn);
    for (int i = 0; i < (int)laserCloudInfoBuf.size(); i++) {   // as written: the bound shrinks while the loop pops
      laserCloudInfoLast = laserCloudInfoBuf.front();
      laserCloudInfoBuf.pop();
      cloudTimeStamp = laserCloudInfoLast.header.stamp;
      if (std::fabs(now.toSec() - cloudTimeStamp.toSec()) < 0.1) break;
    }
    std::vector<float> corner, surf, full;
    if (!wire::fromROSMsgXYZI(laserCloudInfoLast.extracted_corner, corner) || !wire::fromROSMsgXYZI(laserCloudInfoLast.extracted_surface, surf) ||
        !wire::fromROSMsgXYZI(laserCloudInfoLast.cloud_projected, full))
      return Status::BadFields;
    out.frame = core_.cloudHandler(cloudTimeStamp.toSec(), corner, surf);
    if (out.frame == LidarOdometry::FirstFrame) return Status::FirstFrame;
    // updateTransform :572-585 — RegCloud = FullCloudLast moved by [Rotation | Translation] (float path of pcl::transformPointCloud)
    float T[16] = {0};
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[r * 4 + c] = (float)core_.Rotation[r * 3 + c]; T[r * 4 + 3] = (float)core_.Translation[r]; }
    T[15] = 1.f;
    std::vector<float> reg(full.size());
    if (!full.empty()) check(rolo_transform_cloud(ctx_.get(), full.data(), reg.data(), (int)(full.size() / 4), 4, T), "rolo_transform_cloud");
    // pubMessage :655-697
    out.registration_scan = wire::toROSMsgXYZI(reg.data(), reg.size() / 4, cloudTimeStamp, P.baselinkFrame);
    const auto& L = core_.LaserOdomPose;
    out.laser_pose = wire::PoseStamped();
    out.laser_pose.header.frame_id = P.odometryFrame; out.laser_pose.header.stamp = cloudTimeStamp;
    out.laser_pose.pose.position[0] = L[0]; out.laser_pose.pose.position[1] = L[1]; out.laser_pose.pose.position[2] = L[2];
    wire::createQuaternionFromRPY(L[3], L[4], L[5], out.laser_pose.pose.orientation);
    out.laser_odom_incremental = wire::Odometry();
    out.laser_odom_incremental.header.frame_id = P.odometryFrame; out.laser_odom_incremental.header.stamp = cloudTimeStamp;
    out.laser_odom_incremental.child_frame_id = "lidar_odometry";
    out.laser_odom_incremental.pose = out.laser_pose.pose;
    out.odometry_cloud = laserCloudInfoLast;
    out.odometry_cloud.initialGuessX = L[0]; out.odometry_cloud.initialGuessY = L[1]; out.odometry_cloud.initialGuessZ = L[2];
    out.odometry_cloud.initialGuessRoll = L[3]; out.odometry_cloud.initialGuessPitch = L[4]; out.odometry_cloud.initialGuessYaw = L[5];
    out.odometry_cloud.odomAvailable = 1;
    return Status::Published;
  }
  LidarOdometry& core() { return core_; }
  NodeParams P;
  wire::Time cloudTimeStamp;
  wire::CloudInfoStamp laserCloudInfoLast;
  std::queue<wire::CloudInfoStamp> laserCloudInfoBuf;
private:
  Context& ctx_;
  LidarOdometry core_;
};

}  // namespace ros1
}  // namespace rolo
