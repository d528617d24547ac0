// rolo_ros_nodes.hpp — the message-level halves of the three front-end ROS nodes of sdwyc/ROLO, ROS-free: what each node does
// between "a message arrived" and "publish", on the wire structs of rolo_ros_wire.hpp and the HIP node cores of rolo_nodes_hip.hpp.
// The catkin node sources (ros/*.cpp) only convert between ROS message classes and these structs and own subscribers / publishers.
//
//   rolo::ros1::ImageProjectionNode    cloudHandler / odometryHandler          src/imageProjection.cpp:86-93, 150-176, 179-366, 507-512
//   rolo::ros1::FeatureExtractionNode  laserCloudInfoHandler                   src/featureExtraction.cpp:42-49, 71-85, 268-287
//   rolo::ros1::LidarOdometryNode      odometryHandler / cloudHandler / pubMessage   src/lidarOdometry.cpp:394-405, 440-446, 503-570, 655-697
//   rolo::ros1::TransformFusionNode    mapping / lidar odometry handlers, 20 Hz fusion timer, 30 Hz predict timer   src/lidarOdometry.cpp:47-323
//   rolo::ros1::FusedFrontEndNode      the three front-end nodes in ONE process: raw cloud in -> odometry out, clouds stay in HBM
//
// Where the reference calls ros::shutdown() (non-dense cloud, unknown sensor) the handlers return a Status and leave the policy to
// the node source. Deviation, stated: a cloud without a `ring` field is refused (the reference would silently put every point in row 0).
#pragma once
#include <cmath>
#include <deque>
#include <queue>
#include <string>
#include <vector>

#include "rolo_fusion.h"
#include "rolo_nodes_hip.hpp"
#include "rolo_ros_wire.hpp"

namespace rolo {
namespace ros1 {

enum class LidarType { VELODYNE = 0, OUSTER = 1 };

// the keys ParamLoader reads for these three nodes (include/rolo/utility.h:267-333), with its defaults
struct NodeParams {
  std::string pointCloudTopic = "points_raw", odomTopic = "odometry/imu";
  std::string lidarFrame = "base_link", baselinkFrame = "base_link", odometryFrame = "odom", mapFrame = "map";
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


// TransformFusion (src/lidarOdometry.cpp:47-323), the second half of the rolo_lidarOdometry process: smooths the front end's
// odomTopic + "_incremental" poses with the 18-dof ESKF and re-bases them on the back end's last rolo/mapping/odometry. The filter,
// the queue and the float Affine3f chain are rolo_fusion_* (include/rolo_fusion.h, in librolo_hip.so); this class adds what is message-level:
// the odometry template (:124, :203), the 1 s path (:224-240), frames and stamps. `now` = ros::Time::now() of the timer callback.
class TransformFusionNode {
public:
  struct FusionOutputs {   // what fusionTimerHandler publishes (:138-241)
    wire::Odometry odometry;              // odomTopic
    wire::Float32 speed;                  // odomTopic + "/speed"
    bool path_updated = false;            // a pose went onto lidarPath: publish `path` on rolo/lidar_odometry/path if it has subscribers (:233-238)
    wire::Pose odom_to_lidar;             // TF odometryFrame -> baselinkFrame BEFORE the static lidar2Baselink factor (:213-219: the node source multiplies it in)
  };
  struct PredictOutputs {  // what predictTimerHandler publishes (:243-322)
    wire::Header header;                              // stamp = now, frame_id = lidarFrame: of future_path and of future_pose_lidar
    std::vector<rolo_future_point> points;            // autoware_rviz_msgs/Path::points (an external package: filled by field name in the node source)
    wire::PoseWithCovarianceStamped future_pose_lidar;   // the last point, covariance zero
  };
  explicit TransformFusionNode(const NodeParams& p, const rolo_eskf_options* opt = nullptr) : P(p) {
    if (rolo_fusion_create(opt, &f_) != 0) throw Error(-6, "rolo_fusion_create");
  }
  ~TransformFusionNode() { rolo_fusion_destroy(f_); }
  TransformFusionNode(const TransformFusionNode&) = delete;
  TransformFusionNode& operator=(const TransformFusionNode&) = delete;

  // mappingOdometryHandler :109-117 — rolo/mapping/odometry
  void mappingOdometryHandler(const wire::Odometry& odomMsg) { rolo_fusion_mapping_odometry(f_, odomMsg.header.stamp.toSec(), odomMsg.pose.position, odomMsg.pose.orientation); }
  // lidarOdometryHandler :119-125 — odomTopic + "_incremental"
  void lidarOdometryHandler(const wire::Odometry& odomMsg) {
    rolo_fusion_lidar_odometry(f_, odomMsg.header.stamp.toSec(), odomMsg.pose.position, odomMsg.pose.orientation);
    latestLidarOdomTemplate = odomMsg;
  }
  // fusionTimerHandler :138-241. (The map -> odom identity TF of :143-144 is sent by the node source on every tick, published or not.)
  // false: the handler returned early, nothing is published.
  bool fusionTimerHandler(const wire::Time& now, FusionOutputs& out) {
    rolo_fusion_odometry o;
    if (rolo_fusion_timer(f_, now.toSec(), &o) != 1) return false;
    out.odometry = latestLidarOdomTemplate;   // twist.angular and both covariances keep what the template carried
    out.odometry.header.stamp = now; out.odometry.header.frame_id = P.odometryFrame; out.odometry.child_frame_id = P.baselinkFrame;
    for (int i = 0; i < 3; i++) { out.odometry.pose.position[i] = o.position[i]; out.odometry.twist_linear[i] = o.velocity[i]; }
    for (int i = 0; i < 4; i++) out.odometry.pose.orientation[i] = o.orientation[i];
    out.odom_to_lidar = out.odometry.pose;
    out.path_updated = o.path_appended != 0;
    if (out.path_updated) {
      wire::PoseStamped ps; ps.header.stamp = now; ps.header.frame_id = P.odometryFrame; ps.pose = out.odometry.pose;
      path.poses.push_back(ps);
      const double t = now.toSec();
      while (!path.poses.empty() && path.poses.front().header.stamp.toSec() < t - 1.0) path.poses.erase(path.poses.begin());
      path.header.stamp = now; path.header.frame_id = P.odometryFrame;   // (the reference stamps it only when somebody listens: same bytes on the wire)
    }
    out.speed.data = (float)o.speed;
    return true;
  }
  // predictTimerHandler :243-322; false: nothing is published
  bool predictTimerHandler(const wire::Time& now, PredictOutputs& out) {
    const int n = rolo_fusion_predict_timer(f_, nullptr, 0);
    if (n <= 0) return false;
    out.points.resize((size_t)n);
    rolo_fusion_predict_timer(f_, out.points.data(), n);
    out.header = wire::Header(); out.header.stamp = now; out.header.frame_id = P.lidarFrame;
    out.future_pose_lidar = wire::PoseWithCovarianceStamped();
    out.future_pose_lidar.header = out.header;
    const rolo_future_point& last = out.points.back();
    for (int i = 0; i < 3; i++) out.future_pose_lidar.pose.position[i] = last.position[i];
    for (int i = 0; i < 4; i++) out.future_pose_lidar.pose.orientation[i] = last.orientation[i];
    return true;
  }
  rolo_fusion* fusion() { return f_; }
  NodeParams P;
  wire::Odometry latestLidarOdomTemplate;
  wire::Path path;   // lidarPath
private:
  rolo_fusion* f_ = nullptr;
};

// SURVEY 8f.2 — "keep the three node names / topics but run K1 -> K13 in one process with device-resident buffers": ImageProjection's
// cachePointCloud queue (processing starts at the third message, imageProjection.cpp:183) in front of rolo_odom_submit_msg / _collect, which
// unpack the sensor_msgs/PointCloud2 payload on the device and keep range image, feature clouds and both registration inputs in HBM.
// What goes out is what the rolo_lidarOdometry node publishes (odomTopic + "_incremental", pose, odomTopic + "/cloud_info" with the
// feature clouds of the frame; the projected cloud and the index arrays are NOT carried — no consumer of odomTopic + "/cloud_info" reads them
// except the registration_scan debug topic, which this node does not offer). One cloud is in flight while the next one's K1-K4 run.
class FusedFrontEndNode {
public:
  struct Outputs {
    wire::Odometry laser_odom_incremental;
    wire::PoseStamped laser_pose;
    wire::CloudInfoStamp odometry_cloud;
    LidarOdometry::Status frame = LidarOdometry::FirstFrame;
  };
  FusedFrontEndNode(Context& ctx, const NodeParams& p) : P(p), fp_(p.front()), core_(ctx, p.CT_lambda) {}
  void odometryHandler(const wire::Odometry& mappedOdom) { core_.odometryHandler(mappedOdom.header.stamp.toSec()); }

  // cloudHandler of ImageProjection (:158-176) through to pubMessage of LidarOdometry (:655-697) for the cloud that leaves the 3-deep queue
  Status cloudHandler(const wire::PointCloud2& laserCloudMsg, Outputs& out) {
    cloudQueue.push_back(laserCloudMsg);
    if (cloudQueue.size() <= 2) return Status::Queued;
    cur_ = std::move(cloudQueue.front());
    cloudQueue.pop_front();
    const wire::PointCloud2& m = cur_;
    if (P.sensor != LidarType::VELODYNE && P.sensor != LidarType::OUSTER) return Status::BadSensor;
    const std::string timeField = P.sensor == LidarType::OUSTER ? "t" : "time";
    const wire::PointField *fx = m.field("x"), *fy = m.field("y"), *fz = m.field("z"), *fring = m.field("ring"), *ftime = m.field(timeField);
    const size_t n = m.size();
    if (!fx || !fy || !fz || !fring || fx->datatype != wire::PointField::FLOAT32 || fy->datatype != wire::PointField::FLOAT32 ||
        fz->datatype != wire::PointField::FLOAT32 || m.point_step == 0 || m.data.size() < n * (size_t)m.point_step)
      return Status::BadFields;
    if (!m.is_dense) return Status::NonDense;
    rolo_cloud_layout L;
    L.point_step = (int)m.point_step; L.off_x = (int)fx->offset; L.off_y = (int)fy->offset; L.off_z = (int)fz->offset; L.off_ring = (int)fring->offset;
    // the device unpacker reads the drivers' native layouts: ring UINT16 (Velodyne) / UINT8 (Ouster), time FLOAT32 "time" / UINT32 "t"
    if (fring->datatype == wire::PointField::UINT16) L.ring_bytes = 2; else if (fring->datatype == wire::PointField::UINT8) L.ring_bytes = 1; else return Status::BadFields;
    L.off_time = 0; L.time_kind = 0;
    if (ftime && P.sensor == LidarType::VELODYNE && ftime->datatype == wire::PointField::FLOAT32) { L.off_time = (int)ftime->offset; L.time_kind = 1; }
    if (ftime && P.sensor == LidarType::OUSTER && ftime->datatype == wire::PointField::UINT32) { L.off_time = (int)ftime->offset; L.time_kind = 2; }
    const wire::Time stamp = m.header.stamp;
    // deskewCloudInfo :266-366 on the odometry this node published itself (ImageProjection listens to odomTopic + "_incremental")
    // odomAvailable is STICKY in the reference (imageProjection.cpp:150-155: set once two messages have arrived, never cleared), also after the gate loop
    // below has popped the queue under two entries — e.g. across a pause in the stamps; ImageProjectionNode keeps the same flag
    if (P.deskewEnabled && odomAvailable && n > 0) {
      const double timeScanCur = stamp.toSec(), gate = L.time_kind == 0 ? 0.25 : 0.3;
      while (!odomQueue.empty()) { if (std::fabs(timeScanCur - odomQueue.front().header.stamp.toSec()) > gate) odomQueue.pop_front(); else break; }
      if (!odomQueue.empty()) {
        float front6[6], back6[6], incre6[6];
        odom2pose(odomQueue.front(), front6); odom2pose(odomQueue.back(), back6);
        rolo_odom_increment(front6, back6, incre6);
        const double odomTimeDiff = odomQueue.back().header.stamp.toSec() - odomQueue.front().header.stamp.toSec();
        if (odomTimeDiff != 0.0) core_.setDeskew(incre6 + 3, 0.1f, odomTimeDiff, nullptr, 0);   // per-point times: the message's time field, else the azimuth (on the device)
      }
    }
    core_.submitMsg(fp_, stamp.toSec(), m.data.data(), L, (int)n);
    out.frame = core_.collect();
    if (out.frame == LidarOdometry::FirstFrame) return Status::FirstFrame;
    const auto& Lp = core_.LaserOdomPose;
    out.laser_pose = wire::PoseStamped();
    out.laser_pose.header.frame_id = P.odometryFrame; out.laser_pose.header.stamp = stamp;
    out.laser_pose.pose.position[0] = Lp[0]; out.laser_pose.pose.position[1] = Lp[1]; out.laser_pose.pose.position[2] = Lp[2];
    wire::createQuaternionFromRPY(Lp[3], Lp[4], Lp[5], out.laser_pose.pose.orientation);
    out.laser_odom_incremental = wire::Odometry();
    out.laser_odom_incremental.header = out.laser_pose.header;
    out.laser_odom_incremental.child_frame_id = "lidar_odometry";
    out.laser_odom_incremental.pose = out.laser_pose.pose;
    std::vector<float> feat; int nc = 0, ns = 0;
    core_.getFeatures(feat, nc, ns);
    out.odometry_cloud = wire::CloudInfoStamp();
    out.odometry_cloud.header = m.header;
    out.odometry_cloud.extracted_corner = wire::toROSMsgXYZI(feat.data(), (size_t)nc, stamp, P.lidarFrame);
    out.odometry_cloud.extracted_surface = wire::toROSMsgXYZI(feat.data() + 4 * (size_t)nc, (size_t)ns, stamp, P.lidarFrame);
    out.odometry_cloud.extracted_normal = wire::toROSMsgXYZI(nullptr, 0, stamp, P.lidarFrame);
    out.odometry_cloud.cloud_projected = wire::toROSMsgXYZI(nullptr, 0, stamp, P.lidarFrame);
    out.odometry_cloud.initialGuessX = Lp[0]; out.odometry_cloud.initialGuessY = Lp[1]; out.odometry_cloud.initialGuessZ = Lp[2];
    out.odometry_cloud.initialGuessRoll = Lp[3]; out.odometry_cloud.initialGuessPitch = Lp[4]; out.odometry_cloud.initialGuessYaw = Lp[5];
    out.odometry_cloud.odomAvailable = 1;
    odomQueue.push_back(out.laser_odom_incremental);
    if (odomQueue.size() >= 2) odomAvailable = true;   // ImageProjection::odometryHandler :150-155
    return Status::Published;
  }
  LidarOdometry& core() { return core_; }
  NodeParams P;
  std::deque<wire::PointCloud2> cloudQueue;
  std::deque<wire::Odometry> odomQueue;
  bool odomAvailable = false;
private:
  static void odom2pose(const wire::Odometry& o, float pose6[6]) {
    double r, p, y; wire::getRPY(o.pose.orientation, r, p, y);
    pose6[0] = (float)o.pose.position[0]; pose6[1] = (float)o.pose.position[1]; pose6[2] = (float)o.pose.position[2];
    pose6[3] = (float)r; pose6[4] = (float)p; pose6[5] = (float)y;
  }
  FrontParams fp_;
  LidarOdometry core_;
  wire::PointCloud2 cur_;
};

}  // namespace ros1
}  // namespace rolo
