// rolo_nodes_hip.hpp — header-only C++ host mirrors of the three front-end node cores of sdwyc/ROLO over the C ABI
// of librolo_hip.so (include/rolo_hip.h). Class, method and member names follow the reference so that the node
// sources keep reading the way they do; everything ROS (subscribers, publishers, messages) stays in the node.
//
//   rolo::ImageProjection    projectPointCloud() + cloudExtraction()                 src/imageProjection.cpp:399-505
//   rolo::FeatureExtraction  calculateSmoothness() + markOccludedPoints() + extractFeatures()
//                                                                                     src/featureExtraction.cpp:87-266
//   rolo::LidarOdometry      odometryHandler() / cloudHandler()                       src/lidarOdometry.cpp:440-446, 503-570
//                            + the fused device-resident path frame() / submit() / collect() (rolo_odom_frame & co)
//
// Data are plain arrays with the layout of rolo/CloudInfoStamp (msg/CloudInfoStamp.msg:1-28): int32 startRingIndex /
// endRingIndex / pointColInd, float32 pointRange, feature clouds as n x 4 floats (x, y, z, intensity).
// Error behaviour: the reference nodes log and drop the frame or shut down; these classes throw rolo::Error (a
// std::runtime_error carrying the ROLO_E* code) and leave the policy to the node.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rolo_hip.h"

namespace rolo {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& where) : std::runtime_error(where + ": " + (rolo_last_error() ? rolo_last_error() : "")), code(c) {}
};
inline int check(int rc, const char* where) { if (rc < 0) throw Error(rc, where); return rc; }

// one registration context (one HIP stream) — RAII
class Context {
public:
  explicit Context(int device = 0) { check(rolo_ctx_create(device, &ctx_), "rolo_ctx_create"); }
  ~Context() { rolo_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  rolo_ctx* get() const { return ctx_; }
private:
  rolo_ctx* ctx_ = nullptr;
};

// config/params.yaml:20-36 under the names of include/rolo/utility.h (ParamLoader)
struct FrontParams : rolo_front_params {
  FrontParams() { rolo_front_default_params(this); }
  FrontParams(int N_SCAN, int Horizon_SCAN, int downsampleRate = 1, float lidarMinRange = 2.0f, float lidarMaxRange = 1000.0f,
              float edgeThreshold = 0.8f, float surfThreshold = 0.1f, float odometrySurfLeafSize = 0.4f) {
    n_scan = N_SCAN; horizon_scan = Horizon_SCAN; downsample_rate = downsampleRate; lidar_min_range = lidarMinRange;
    lidar_max_range = lidarMaxRange; edge_threshold = edgeThreshold; surf_threshold = surfThreshold; odometry_surf_leaf_size = odometrySurfLeafSize;
  }
};

// The per-frame arrays ImageProjection publishes in rolo/cloud_info
struct CloudInfo {
  std::vector<int32_t> startRingIndex, endRingIndex, pointColInd;
  std::vector<float> pointRange;
  std::vector<float> extractedCloud;  // cloud_projected: N x 4 (x, y, z, intensity = ring * z, imageProjection.cpp:410)
  int n_valid = 0;
};

class ImageProjection {
public:
  ImageProjection(Context& ctx, const FrontParams& p) : ctx_(ctx), p_(p) {}
  // deskewCloudInfo() + deskewPoint() (:266-396) for the next projectPointCloud: odomIncreRoll/Pitch/Yaw, scanPeriod,
  // odomTimeDiff and rel_time[i] = fabs(point.time) of raw point i (host array of n values)
  void setDeskew(const float odomIncreRPY[3], float scanPeriod, double odomTimeDiff, const float* rel_time, int n, bool deskewEnabled = true) {
    rolo_deskew d{deskewEnabled ? 1 : 0, {odomIncreRPY[0], odomIncreRPY[1], odomIncreRPY[2]}, scanPeriod, odomTimeDiff};
    check(rolo_front_set_deskew(ctx_.get(), &d, rel_time, n, 0), "rolo_front_set_deskew");
  }
  // projectPointCloud() + cloudExtraction(): `pts` = n records of `stride` floats (x, y, z first), `ring` the ring field
  const CloudInfo& projectPointCloud(const float* pts, int stride, const uint16_t* ring, int n) {
    const size_t npix = (size_t)p_.n_scan * p_.horizon_scan;
    info_.startRingIndex.assign(p_.n_scan, 0); info_.endRingIndex.assign(p_.n_scan, 0);
    info_.pointColInd.assign(npix, 0); info_.pointRange.assign(npix, 0.f); info_.extractedCloud.assign(npix * 4, 0.f);
    check(rolo_project_frame(ctx_.get(), &p_, pts, stride, ring, n, info_.extractedCloud.data(), info_.pointColInd.data(), info_.pointRange.data(),
                             info_.startRingIndex.data(), info_.endRingIndex.data(), nullptr, &info_.n_valid), "rolo_project_frame");
    info_.pointColInd.resize(info_.n_valid); info_.pointRange.resize(info_.n_valid); info_.extractedCloud.resize((size_t)info_.n_valid * 4);
    return info_;
  }
  const CloudInfo& cloudInfo() const { return info_; }
private:
  Context& ctx_;
  FrontParams p_;
  CloudInfo info_;
};

class FeatureExtraction {
public:
  FeatureExtraction(Context& ctx, const FrontParams& p) : ctx_(ctx), p_(p) {}
  // runs on the arrays the preceding ImageProjection::projectPointCloud left on the device (same Context)
  void extractFeatures(int n_valid) {
    cornerCloud.assign((size_t)(n_valid > 0 ? n_valid : 1) * 4, 0.f); surfaceCloud.assign((size_t)(n_valid > 0 ? n_valid : 1) * 4, 0.f);
    int nc = 0, ns = 0;
    check(rolo_extract_features(ctx_.get(), &p_, cornerCloud.data(), &nc, surfaceCloud.data(), &ns, nullptr, nullptr, nullptr), "rolo_extract_features");
    cornerCloud.resize((size_t)nc * 4); surfaceCloud.resize((size_t)ns * 4);
  }
  std::vector<float> cornerCloud, surfaceCloud;  // extracted_corner / extracted_surface: n x 4
private:
  Context& ctx_;
  FrontParams p_;
};

class LidarOdometry {
public:
  // status of a frame, as rolo_odom_cloud returns it
  enum Status { FirstFrame = 0, Gated = 1, Registered = 2 };

  explicit LidarOdometry(Context& ctx, float CT_lambda = 0.3f, double polar_theta = 0.175, double polar_phi = 0.175, double polar_r = 2.0) : ctx_(ctx) {
    rolo_params P; rolo_default_params(&P);
    P.voxel_type = ROLO_VOXEL_POLAR;  // rot_vgicp.setPolarResolution(0.175, 0.175, 2.0), lidarOdometry.cpp:462
    P.polar_resolution[0] = polar_theta; P.polar_resolution[1] = polar_phi; P.polar_resolution[2] = polar_r;
    check(rolo_set_params(ctx_.get(), &P), "rolo_set_params");
    check(rolo_odom_create(ctx_.get(), CT_lambda, &odom_), "rolo_odom_create");
  }
  ~LidarOdometry() { rolo_odom_destroy(odom_); }
  LidarOdometry(const LidarOdometry&) = delete;
  LidarOdometry& operator=(const LidarOdometry&) = delete;

  void odometryHandler(double stamp) { check(rolo_odom_backend_odometry(odom_, stamp), "rolo_odom_backend_odometry"); }  // :440-446
  // :503-570 between fromROSMsg and pubMessage; clouds n x 4 floats
  Status cloudHandler(double stamp, const std::vector<float>& cornerCloud, const std::vector<float>& surfaceCloud) {
    return (Status)check(rolo_odom_cloud(odom_, stamp, cornerCloud.data(), (int)(cornerCloud.size() / 4), surfaceCloud.data(), (int)(surfaceCloud.size() / 4),
                                         LaserOdomPose.data(), Rotation.data(), Translation.data()), "rolo_odom_cloud");
  }
  // the three node cores fused, device-resident (raw frame -> pose)
  Status frame(const FrontParams& p, double stamp, const float* pts, int stride, const uint16_t* ring, int n, bool on_device = false) {
    return (Status)check(rolo_odom_frame(odom_, &p, stamp, pts, stride, ring, n, on_device ? 1 : 0, LaserOdomPose.data(), Rotation.data(), Translation.data(),
                                         counts.data()), "rolo_odom_frame");
  }
  void submit(const FrontParams& p, double stamp, const float* pts, int stride, const uint16_t* ring, int n, bool on_device = false) {
    check(rolo_odom_submit(odom_, &p, stamp, pts, stride, ring, n, on_device ? 1 : 0), "rolo_odom_submit");
  }
  // submit() from the payload of the sensor_msgs/PointCloud2 itself (field offsets from msg.fields): no host-side extraction
  void submitMsg(const FrontParams& p, double stamp, const uint8_t* data, const rolo_cloud_layout& layout, int n_points, bool on_device = false) {
    check(rolo_odom_submit_msg(odom_, &p, stamp, data, &layout, n_points, on_device ? 1 : 0), "rolo_odom_submit_msg");
  }
  Status collect() { return (Status)check(rolo_odom_collect(odom_, LaserOdomPose.data(), Rotation.data(), Translation.data(), counts.data()), "rolo_odom_collect"); }
  // de-skew of the next submit() / frame() (see ImageProjection::setDeskew)
  void setDeskew(const float odomIncreRPY[3], float scanPeriod, double odomTimeDiff, const float* rel_time, int n, bool deskewEnabled = true) {
    rolo_deskew d{deskewEnabled ? 1 : 0, {odomIncreRPY[0], odomIncreRPY[1], odomIncreRPY[2]}, scanPeriod, odomTimeDiff};
    check(rolo_odom_set_deskew(odom_, &d, rel_time, n, 0), "rolo_odom_set_deskew");
  }
  // extracted_corner ++ extracted_surface of the last collected fused frame (n x 4 floats, corners first)
  void getFeatures(std::vector<float>& features, int& n_corner, int& n_surface) {
    check(rolo_odom_get_features(odom_, nullptr, 0, &n_corner, &n_surface), "rolo_odom_get_features");
    features.assign((size_t)(n_corner + n_surface > 0 ? n_corner + n_surface : 1) * 4, 0.f);
    check(rolo_odom_get_features(odom_, features.data(), n_corner + n_surface, &n_corner, &n_surface), "rolo_odom_get_features");
  }
  void setReuseCovariances(bool on) { check(rolo_odom_set_option(odom_, ROLO_ODOM_REUSE_COVARIANCES, on ? 1 : 0), "rolo_odom_set_option"); }

  std::array<float, 6> LaserOdomPose{};   // x, y, z, roll, pitch, yaw — what pubMessage publishes (:680-684)
  std::array<double, 9> Rotation{};       // frame-to-frame step, row-major
  std::array<double, 3> Translation{};
  std::array<int, 3> counts{};            // N, n_corner, n_surface of the last fused frame
private:
  Context& ctx_;
  rolo_odom* odom_ = nullptr;
};

}  // namespace rolo
