// rolo_ros_wire.hpp — the messages of the ROLO front-end topic surface as plain C++ structs with their ROS1 wire
// (de)serialisation, without ROS. Lets the node cores (rolo_ros_nodes.hpp) be built and byte-tested where no ROS exists;
// under catkin, ros/*.cpp copy field by field between these structs and the generated message classes.
//
//   std_msgs/Header, sensor_msgs/PointField, sensor_msgs/PointCloud2 (in:  rolo/pointCloudTopic, imageProjection.cpp:89)
//   rolo/CloudInfoStamp                       (msg/CloudInfoStamp.msg:1-28; rolo/cloud_info, rolo/feature/cloud_info, odomTopic/cloud_info)
//   nav_msgs/Odometry, geometry_msgs/PoseStamped (out: odomTopic_incremental [/pose], lidarOdometry.cpp:655-684; in: rolo/mapping/odometry)
//
// ROS1 serialisation (roscpp_serialization): little-endian fixed-width scalars in declaration order, bool = uint8, time = uint32 sec +
// uint32 nsec, string = uint32 length + bytes (no terminator), T[] = uint32 count + elements, T[N] = elements only.
// Also: pcl::toROSMsg / fromROSMsg for pcl::PointXYZI (32-byte records: x@0 y@4 z@8 data[3]=1 intensity@16),
// tf::createQuaternionFromRPY and tf::Matrix3x3(q).getRPY restated (ROS tf is not part of the reference tree).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rolo {
namespace wire {

struct Time { uint32_t sec = 0, nsec = 0; double toSec() const { return (double)sec + 1e-9 * (double)nsec; } };
inline Time timeFromSec(double t) {   // ros::Time::fromSec
  Time o; const double fl = std::floor(t); o.sec = (uint32_t)fl; o.nsec = (uint32_t)std::lround((t - fl) * 1e9);
  if (o.nsec >= 1000000000u) { o.sec += 1; o.nsec -= 1000000000u; }
  return o;
}
struct Header { uint32_t seq = 0; Time stamp; std::string frame_id; };

struct PointField {   // sensor_msgs/PointField
  enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0;
};
struct PointCloud2 {
  Header header; uint32_t height = 0, width = 0; std::vector<PointField> fields; uint8_t is_bigendian = 0;
  uint32_t point_step = 0, row_step = 0; std::vector<uint8_t> data; uint8_t is_dense = 0;
  size_t size() const { return (size_t)height * width; }
  const PointField* field(const std::string& name) const { for (const auto& f : fields) if (f.name == name) return &f; return nullptr; }
};

struct CloudInfoStamp {   // msg/CloudInfoStamp.msg:1-28, same order
  Header header;
  std::vector<int32_t> startRingIndex, endRingIndex, pointColInd;
  std::vector<float> pointRange;
  float startOrientation = 0, endOrientation = 0, orientationDiff = 0;
  float initialGuessX = 0, initialGuessY = 0, initialGuessZ = 0, initialGuessRoll = 0, initialGuessPitch = 0, initialGuessYaw = 0;
  std::vector<float> covariance;
  uint8_t odomAvailable = 0;
  PointCloud2 cloud_projected, extracted_corner, extracted_surface, extracted_normal, extracted_ground;
};

struct Pose { double position[3] = {0, 0, 0}; double orientation[4] = {0, 0, 0, 0}; /* x, y, z, w */ };
struct PoseStamped { Header header; Pose pose; };
struct Path { Header header; std::vector<PoseStamped> poses; };                                          // nav_msgs/Path
struct Float32 { float data = 0.f; };                                                                   // std_msgs/Float32
struct PoseWithCovarianceStamped { Header header; Pose pose; double covariance[36] = {0}; };            // geometry_msgs/PoseWithCovarianceStamped
struct Odometry {   // nav_msgs/Odometry
  Header header; std::string child_frame_id; Pose pose; double pose_covariance[36] = {0};
  double twist_linear[3] = {0, 0, 0}, twist_angular[3] = {0, 0, 0}; double twist_covariance[36] = {0};
};

// ---- writer / reader ---------------------------------------------------------------------------------------------------------
class Writer {
public:
  explicit Writer(std::vector<uint8_t>& out) : o_(out) {}
  template <typename T> void pod(T v) { uint8_t b[sizeof(T)]; std::memcpy(b, &v, sizeof(T)); o_.insert(o_.end(), b, b + sizeof(T)); }   // little-endian host (x86-64)
  void str(const std::string& s) { pod<uint32_t>((uint32_t)s.size()); o_.insert(o_.end(), s.begin(), s.end()); }
  template <typename T> void arr(const std::vector<T>& v) { pod<uint32_t>((uint32_t)v.size()); raw(v.data(), v.size() * sizeof(T)); }
  void raw(const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); o_.insert(o_.end(), b, b + n); }
private:
  std::vector<uint8_t>& o_;
};
class Reader {
public:
  Reader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  bool ok() const { return ok_; }
  size_t consumed() const { return i_; }
  template <typename T> T pod() { T v{}; if (!need(sizeof(T))) return v; std::memcpy(&v, p_ + i_, sizeof(T)); i_ += sizeof(T); return v; }
  std::string str() { const uint32_t len = pod<uint32_t>(); if (!need(len)) return {}; std::string s((const char*)p_ + i_, len); i_ += len; return s; }
  template <typename T> void arr(std::vector<T>& v) {
    const uint32_t cnt = pod<uint32_t>();
    if (!ok_ || cnt > (n_ - i_) / sizeof(T)) { ok_ = false; v.clear(); return; }
    v.resize(cnt); raw(v.data(), (size_t)cnt * sizeof(T));
  }
  void raw(void* dst, size_t n) { if (!need(n)) return; if (n) std::memcpy(dst, p_ + i_, n); i_ += n; }
private:
  bool need(size_t k) { if (!ok_ || k > n_ - i_) { ok_ = false; return false; } return true; }
  const uint8_t* p_; size_t n_, i_ = 0; bool ok_ = true;
};

inline void write(Writer& w, const Header& h) { w.pod(h.seq); w.pod(h.stamp.sec); w.pod(h.stamp.nsec); w.str(h.frame_id); }
inline void read(Reader& r, Header& h) { h.seq = r.pod<uint32_t>(); h.stamp.sec = r.pod<uint32_t>(); h.stamp.nsec = r.pod<uint32_t>(); h.frame_id = r.str(); }

inline void write(Writer& w, const PointCloud2& m) {
  write(w, m.header); w.pod(m.height); w.pod(m.width);
  w.pod<uint32_t>((uint32_t)m.fields.size());
  for (const auto& f : m.fields) { w.str(f.name); w.pod(f.offset); w.pod(f.datatype); w.pod(f.count); }
  w.pod(m.is_bigendian); w.pod(m.point_step); w.pod(m.row_step); w.arr(m.data); w.pod(m.is_dense);
}
inline void read(Reader& r, PointCloud2& m) {
  read(r, m.header); m.height = r.pod<uint32_t>(); m.width = r.pod<uint32_t>();
  const uint32_t nf = r.pod<uint32_t>();
  m.fields.clear();
  for (uint32_t i = 0; i < nf && r.ok(); i++) { PointField f; f.name = r.str(); f.offset = r.pod<uint32_t>(); f.datatype = r.pod<uint8_t>(); f.count = r.pod<uint32_t>(); m.fields.push_back(f); }
  m.is_bigendian = r.pod<uint8_t>(); m.point_step = r.pod<uint32_t>(); m.row_step = r.pod<uint32_t>(); r.arr(m.data); m.is_dense = r.pod<uint8_t>();
}

inline void write(Writer& w, const CloudInfoStamp& m) {
  write(w, m.header); w.arr(m.startRingIndex); w.arr(m.endRingIndex); w.arr(m.pointColInd); w.arr(m.pointRange);
  w.pod(m.startOrientation); w.pod(m.endOrientation); w.pod(m.orientationDiff);
  w.pod(m.initialGuessX); w.pod(m.initialGuessY); w.pod(m.initialGuessZ); w.pod(m.initialGuessRoll); w.pod(m.initialGuessPitch); w.pod(m.initialGuessYaw);
  w.arr(m.covariance); w.pod(m.odomAvailable);
  write(w, m.cloud_projected); write(w, m.extracted_corner); write(w, m.extracted_surface); write(w, m.extracted_normal); write(w, m.extracted_ground);
}
inline void read(Reader& r, CloudInfoStamp& m) {
  read(r, m.header); r.arr(m.startRingIndex); r.arr(m.endRingIndex); r.arr(m.pointColInd); r.arr(m.pointRange);
  m.startOrientation = r.pod<float>(); m.endOrientation = r.pod<float>(); m.orientationDiff = r.pod<float>();
  m.initialGuessX = r.pod<float>(); m.initialGuessY = r.pod<float>(); m.initialGuessZ = r.pod<float>();
  m.initialGuessRoll = r.pod<float>(); m.initialGuessPitch = r.pod<float>(); m.initialGuessYaw = r.pod<float>();
  r.arr(m.covariance); m.odomAvailable = r.pod<uint8_t>();
  read(r, m.cloud_projected); read(r, m.extracted_corner); read(r, m.extracted_surface); read(r, m.extracted_normal); read(r, m.extracted_ground);
}

inline void write(Writer& w, const Pose& p) { w.raw(p.position, sizeof(p.position)); w.raw(p.orientation, sizeof(p.orientation)); }
inline void read(Reader& r, Pose& p) { r.raw(p.position, sizeof(p.position)); r.raw(p.orientation, sizeof(p.orientation)); }
inline void write(Writer& w, const PoseStamped& m) { write(w, m.header); write(w, m.pose); }
inline void read(Reader& r, PoseStamped& m) { read(r, m.header); read(r, m.pose); }
inline void write(Writer& w, const Odometry& m) {
  write(w, m.header); w.str(m.child_frame_id); write(w, m.pose); w.raw(m.pose_covariance, sizeof(m.pose_covariance));
  w.raw(m.twist_linear, sizeof(m.twist_linear)); w.raw(m.twist_angular, sizeof(m.twist_angular)); w.raw(m.twist_covariance, sizeof(m.twist_covariance));
}
inline void read(Reader& r, Odometry& m) {
  read(r, m.header); m.child_frame_id = r.str(); read(r, m.pose); r.raw(m.pose_covariance, sizeof(m.pose_covariance));
  r.raw(m.twist_linear, sizeof(m.twist_linear)); r.raw(m.twist_angular, sizeof(m.twist_angular)); r.raw(m.twist_covariance, sizeof(m.twist_covariance));
}

inline void write(Writer& w, const Path& m) { write(w, m.header); w.pod<uint32_t>((uint32_t)m.poses.size()); for (const auto& p : m.poses) write(w, p); }
inline void read(Reader& r, Path& m) {
  read(r, m.header);
  const uint32_t cnt = r.pod<uint32_t>();
  m.poses.clear();
  for (uint32_t i = 0; i < cnt && r.ok(); i++) { PoseStamped p; read(r, p); if (r.ok()) m.poses.push_back(p); }
}
inline void write(Writer& w, const Float32& m) { w.pod(m.data); }
inline void read(Reader& r, Float32& m) { m.data = r.pod<float>(); }
inline void write(Writer& w, const PoseWithCovarianceStamped& m) { write(w, m.header); write(w, m.pose); w.raw(m.covariance, sizeof(m.covariance)); }
inline void read(Reader& r, PoseWithCovarianceStamped& m) { read(r, m.header); read(r, m.pose); r.raw(m.covariance, sizeof(m.covariance)); }

template <typename M> std::vector<uint8_t> serialize(const M& m) { std::vector<uint8_t> out; Writer w(out); write(w, m); return out; }
// true iff the whole buffer is one well-formed message
template <typename M> bool deserialize(const uint8_t* p, size_t n, M& m) { Reader r(p, n); read(r, m); return r.ok() && r.consumed() == n; }

// ---- pcl::toROSMsg / pcl::fromROSMsg for pcl::PointXYZI ------------------------------------------------------------------------
// `pts` = n x 4 floats (x, y, z, intensity). Record = the 32-byte pcl::PointXYZI: data[4] = (x, y, z, 1), intensity, 12 bytes padding.
inline PointCloud2 toROSMsgXYZI(const float* pts, size_t n, const Time& stamp, const std::string& frame) {
  PointCloud2 m;
  m.header.stamp = stamp; m.header.frame_id = frame;
  m.height = 1; m.width = (uint32_t)n; m.is_bigendian = 0; m.point_step = 32; m.row_step = (uint32_t)(32 * n); m.is_dense = 1;
  const char* names[4] = {"x", "y", "z", "intensity"}; const uint32_t offs[4] = {0, 4, 8, 16};
  for (int i = 0; i < 4; i++) { PointField f; f.name = names[i]; f.offset = offs[i]; f.datatype = PointField::FLOAT32; f.count = 1; m.fields.push_back(f); }
  m.data.assign(32 * n, 0);
  for (size_t i = 0; i < n; i++) {
    float rec[8] = {pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], 1.0f, pts[4 * i + 3], 0.f, 0.f, 0.f};
    std::memcpy(m.data.data() + 32 * i, rec, 32);
  }
  return m;
}
// fields are matched by name and must be FLOAT32 (pcl::fromROSMsg's field map); a missing field leaves the PCL default (0; data[3] = 1)
inline bool fromROSMsgXYZI(const PointCloud2& m, std::vector<float>& pts /* n x 4 */) {
  const size_t n = m.size();
  pts.assign(n * 4, 0.f);
  if (m.data.size() < (size_t)m.point_step * n) return false;
  const char* names[4] = {"x", "y", "z", "intensity"};
  for (int c = 0; c < 4; c++) {
    const PointField* f = m.field(names[c]);
    if (!f) continue;
    if (f->datatype != PointField::FLOAT32 || f->offset + 4 > m.point_step) return false;
    for (size_t i = 0; i < n; i++) std::memcpy(&pts[4 * i + c], m.data.data() + (size_t)m.point_step * i + f->offset, 4);
  }
  return true;
}

// ---- tf ------------------------------------------------------------------------------------------------------------------------
// tf::createQuaternionFromRPY -> tf::Quaternion::setRPY (tfScalar = double): q = (x, y, z, w)
inline void createQuaternionFromRPY(double roll, double pitch, double yaw, double q[4]) {
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}
// tf::Matrix3x3(q).getRPY(roll, pitch, yaw): setRotation(q) then getEulerYPR (solution 1)
inline void getRPY(const double q[4], double& roll, double& pitch, double& yaw) {
  const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double s = 2.0 / d;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs, xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs, yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  if (std::fabs(m20) >= 1.0) {   // gimbal lock branch of getEulerYPR
    yaw = 0.0;
    const double delta = std::atan2(m21, m22);
    if (m20 < 0) { pitch = M_PI / 2.0; roll = delta; } else { pitch = -M_PI / 2.0; roll = delta; }
    return;
  }
  pitch = -std::asin(m20);
  const double c = std::cos(pitch);
  roll = std::atan2(m21 / c, m22 / c);
  yaw = std::atan2(m10 / c, m00 / c);
}

}  // namespace wire
}  // namespace rolo
