// Byte-level check of the ROS face without ROS (include/rolo_ros_wire.hpp, include/rolo_ros_nodes.hpp), in a C++-only process.
//
//   ros_wire_demo roundtrip <pc2|cis|odom|pose> in.bin out.bin     deserialise + re-serialise one ROS1 wire message (no GPU)
//   ros_wire_demo quat roll pitch yaw                               tf::createQuaternionFromRPY, then tf::Matrix3x3(q).getRPY of it (no GPU)
//   ros_wire_demo chain <velodyne|ouster> N_SCAN Horizon_SCAN deskew(0|1) backend_at outdir msg0.bin msg1.bin ...
//       serialized sensor_msgs/PointCloud2 messages through the three nodes, each on its own context, every hop as wire bytes (what
//       TCPROS would carry): writes outdir/cloud_info_<k>.bin, feature_info_<k>.bin, odom_<k>.bin, odom_cloud_<k>.bin, pose_<k>.bin
//       (k = index of the processed cloud) and prints one status line per input message. The back end's first odometry arrives
//       before input message `backend_at`. The process of node C is the whole rolo_lidarOdometry: every odometry it publishes (and the back end's
//       message) also goes to a TransformFusionNode, whose 20 Hz timer is ticked at stamp + 0.02 and + 0.045 of every input message and whose 30 Hz
//       timer once: outdir/fused_<idx>_<j>.bin (odomTopic), speed_<idx>_<j>.bin, path_<idx>_<j>.bin (when the path grew), future_<idx>.bin
//       (future_pose_lidar) + one "fusion" status line per tick.
//   ros_wire_demo fused <velodyne|ouster> N_SCAN Horizon_SCAN deskew(0|1) backend_at outdir msg0.bin msg1.bin ...
//       the same messages through ONE FusedFrontEndNode: writes odom_<k>.bin, pose_<k>.bin, odom_cloud_<k>.bin
//   ros_wire_demo fusion backend.bin outdir mapping_at_csv odom_msgs.bin       (no GPU)
//       odom_msgs.bin = [u32 length][serialized nav_msgs/Odometry] ... through TransformFusionNode alone: message k is handed to
//       lidarOdometryHandler, before the messages listed in mapping_at_csv the message of backend.bin (its stamp replaced by that of message
//       k - 3) to mappingOdometryHandler, then the fusion timer at stamp + 0.02 and + 0.045 and the predict timer; writes outdir/fused.bin =
//       [u32 tick][u32 length][serialized odomTopic message] ..., speed.bin = [u32 tick][f32] ..., path_len.bin, future.bin likewise
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include <sstream>

#include "rolo_ros_nodes.hpp"

using namespace rolo;

static std::vector<uint8_t> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& path, const std::vector<uint8_t>& b) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(b.data()), (std::streamsize)b.size());
}
// ROLO_DEMO_PARAMS="edgeThreshold=1.0,CT_lambda=1.0,...": the hot-path keys of a named reference configuration (config/params_os.yaml, config/M2UD/params.yaml)
// on top of the defaults the modes below set — what ros/rolo_ros_convert.hpp reads from the parameter server
static void apply_param_overrides(ros1::NodeParams& P) {
  const char* e = std::getenv("ROLO_DEMO_PARAMS");
  if (!e) return;
  std::stringstream ss(e); std::string tok;
  while (std::getline(ss, tok, ',')) {
    const size_t q = tok.find('=');
    if (q == std::string::npos) continue;
    const std::string k = tok.substr(0, q); const float v = (float)std::atof(tok.substr(q + 1).c_str());
    if (k == "edgeThreshold") P.edgeThreshold = v; else if (k == "surfThreshold") P.surfThreshold = v; else if (k == "odometrySurfLeafSize") P.odometrySurfLeafSize = v;
    else if (k == "CT_lambda") P.CT_lambda = v; else if (k == "lidarMinRange") P.lidarMinRange = v; else if (k == "lidarMaxRange") P.lidarMaxRange = v;
    else if (k == "downsampleRate") P.downsampleRate = (int)v;
    else { std::fprintf(stderr, "ROLO_DEMO_PARAMS: unknown key %s\n", k.c_str()); std::exit(2); }
  }
}

template <typename M> static int roundtrip(const std::string& in, const std::string& out) {
  const std::vector<uint8_t> b = slurp(in);
  M m;
  if (!wire::deserialize(b.data(), b.size(), m)) { std::printf("malformed\n"); return 3; }
  spit(out, wire::serialize(m));
  std::printf("ok %zu\n", b.size());
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  const std::string mode = argv[1];
  if (mode == "roundtrip" && argc == 5) {
    const std::string t = argv[2];
    if (t == "pc2") return roundtrip<wire::PointCloud2>(argv[3], argv[4]);
    if (t == "cis") return roundtrip<wire::CloudInfoStamp>(argv[3], argv[4]);
    if (t == "odom") return roundtrip<wire::Odometry>(argv[3], argv[4]);
    if (t == "pose") return roundtrip<wire::PoseStamped>(argv[3], argv[4]);
    return 1;
  }
  if (mode == "quat" && argc == 5) {
    double q[4], r, p, y;
    wire::createQuaternionFromRPY(std::atof(argv[2]), std::atof(argv[3]), std::atof(argv[4]), q);
    wire::getRPY(q, r, p, y);
    std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", q[0], q[1], q[2], q[3], r, p, y);
    return 0;
  }
  if (mode == "fusion" && argc == 6) {
    ros1::NodeParams P; P.odomTopic = "odometry/lidar"; P.lidarFrame = "lidar_link"; P.baselinkFrame = "base_link"; P.odometryFrame = "odom";
    const std::vector<uint8_t> be = slurp(argv[2]);
    wire::Odometry backend;
    if (!wire::deserialize(be.data(), be.size(), backend)) return 3;
    const std::string outdir = argv[3];
    std::vector<int> mapping_at;
    { std::stringstream ss(argv[4]); std::string tok; while (std::getline(ss, tok, ',')) if (!tok.empty()) mapping_at.push_back(std::atoi(tok.c_str())); }
    const std::vector<uint8_t> all = slurp(argv[5]);
    std::vector<wire::Odometry> msgs;
    for (size_t i = 0; i + 4 <= all.size();) {
      uint32_t len; std::memcpy(&len, &all[i], 4); i += 4;
      wire::Odometry m;
      if (i + len > all.size() || !wire::deserialize(&all[i], len, m)) { std::printf("malformed odometry %zu\n", msgs.size()); return 3; }
      msgs.push_back(m); i += len;
    }
    ros1::TransformFusionNode TF(P);
    std::vector<uint8_t> fused, speed, plen, future;
    auto put32 = [](std::vector<uint8_t>& o, uint32_t v) { uint8_t b[4]; std::memcpy(b, &v, 4); o.insert(o.end(), b, b + 4); };
    uint32_t tick = 0;
    for (size_t k = 0; k < msgs.size(); k++) {
      for (int at : mapping_at) if (at == (int)k && k >= 3) { wire::Odometry b = backend; b.header.stamp = msgs[k - 3].header.stamp; TF.mappingOdometryHandler(b); }
      TF.lidarOdometryHandler(msgs[k]);
      for (double dt : {0.02, 0.045}) {
        ros1::TransformFusionNode::FusionOutputs o;
        const wire::Time now = wire::timeFromSec(msgs[k].header.stamp.toSec() + dt);
        if (TF.fusionTimerHandler(now, o)) {
          const std::vector<uint8_t> b = wire::serialize(o.odometry);
          put32(fused, tick); put32(fused, (uint32_t)b.size()); fused.insert(fused.end(), b.begin(), b.end());
          const std::vector<uint8_t> sb = wire::serialize(o.speed);
          put32(speed, tick); speed.insert(speed.end(), sb.begin(), sb.end());
          put32(plen, tick); put32(plen, (uint32_t)TF.path.poses.size() | (o.path_updated ? 0x80000000u : 0u));
        }
        tick++;
      }
      ros1::TransformFusionNode::PredictOutputs po;
      if (TF.predictTimerHandler(msgs[k].header.stamp, po)) {
        const std::vector<uint8_t> b = wire::serialize(po.future_pose_lidar);
        put32(future, (uint32_t)k); put32(future, (uint32_t)po.points.size()); put32(future, (uint32_t)b.size()); future.insert(future.end(), b.begin(), b.end());
      }
    }
    spit(outdir + "/fused.bin", fused); spit(outdir + "/speed.bin", speed); spit(outdir + "/path_len.bin", plen); spit(outdir + "/future.bin", future);
    std::printf("ok %zu messages %u ticks\n", msgs.size(), tick);
    return 0;
  }
  if (mode == "fused" && argc >= 9) {
    ros1::NodeParams P;
    P.sensor = std::string(argv[2]) == "ouster" ? ros1::LidarType::OUSTER : ros1::LidarType::VELODYNE;
    P.N_SCAN = std::atoi(argv[3]); P.Horizon_SCAN = std::atoi(argv[4]);
    P.deskewEnabled = std::atoi(argv[5]) != 0;
    const int backend_at = std::atoi(argv[6]);
    const std::string outdir = argv[7];
    P.lidarMinRange = 2.0f; P.edgeThreshold = 0.8f; P.surfThreshold = 0.1f; P.odometrySurfLeafSize = 0.4f; P.CT_lambda = 0.3f;
    P.odomTopic = "odometry/lidar"; P.lidarFrame = "lidar_link"; P.baselinkFrame = "base_link"; P.odometryFrame = "odom";
    try {
      Context ctx;
      ros1::FusedFrontEndNode F(ctx, P);
      int processed = 0;
      for (int k = 8; k < argc; k++) {
        const int idx = k - 8;
        const std::vector<uint8_t> raw = slurp(argv[k]);
        wire::PointCloud2 msg;
        if (!wire::deserialize(raw.data(), raw.size(), msg)) { std::printf("msg %d malformed\n", idx); return 3; }
        if (idx == backend_at) { wire::Odometry mapped; mapped.header.stamp = msg.header.stamp; F.odometryHandler(mapped); }
        ros1::FusedFrontEndNode::Outputs o;
        const ros1::Status st = F.cloudHandler(msg, o);
        if (st == ros1::Status::Queued) { std::printf("msg %d queued\n", idx); continue; }
        if (st == ros1::Status::Published) {
          spit(outdir + "/odom_" + std::to_string(processed) + ".bin", wire::serialize(o.laser_odom_incremental));
          spit(outdir + "/pose_" + std::to_string(processed) + ".bin", wire::serialize(o.laser_pose));
          spit(outdir + "/odom_cloud_" + std::to_string(processed) + ".bin", wire::serialize(o.odometry_cloud));
        }
        std::printf("msg %d cloud %d fused %d frame %d\n", idx, processed, (int)st, (int)o.frame);
        processed++;
      }
    } catch (const rolo::Error& e) {
      std::fprintf(stderr, "rolo::Error %d: %s\n", e.code, e.what());
      return 4;
    }
    return 0;
  }
  if (mode == "chain" && argc >= 9) {
    ros1::NodeParams P;
    P.sensor = std::string(argv[2]) == "ouster" ? ros1::LidarType::OUSTER : ros1::LidarType::VELODYNE;
    P.N_SCAN = std::atoi(argv[3]); P.Horizon_SCAN = std::atoi(argv[4]);
    P.deskewEnabled = std::atoi(argv[5]) != 0;
    const int backend_at = std::atoi(argv[6]);
    const std::string outdir = argv[7];
    // config/params.yaml values of the shipped configuration
    P.lidarMinRange = 2.0f; P.edgeThreshold = 0.8f; P.surfThreshold = 0.1f; P.odometrySurfLeafSize = 0.4f; P.CT_lambda = 0.3f;
    apply_param_overrides(P);
    P.odomTopic = "odometry/lidar"; P.lidarFrame = "lidar_link"; P.baselinkFrame = "base_link"; P.odometryFrame = "odom";
    try {
      Context ctxA, ctxB, ctxC;   // three nodes = three processes in the reference: nothing is shared but the messages
      ros1::ImageProjectionNode A(ctxA, P);
      ros1::FeatureExtractionNode B(ctxB, P);
      ros1::LidarOdometryNode C(ctxC, P);
      ros1::TransformFusionNode TF(P);   // the other object of the rolo_lidarOdometry process (lidarOdometry.cpp:720-721)
      auto tick = [&](int idx, const wire::Time& stamp) {
        int j = 0;
        for (double dt : {0.02, 0.045}) {
          ros1::TransformFusionNode::FusionOutputs fo;
          const bool pub = TF.fusionTimerHandler(wire::timeFromSec(stamp.toSec() + dt), fo);
          if (pub) {
            spit(outdir + "/fused_" + std::to_string(idx) + "_" + std::to_string(j) + ".bin", wire::serialize(fo.odometry));
            spit(outdir + "/speed_" + std::to_string(idx) + "_" + std::to_string(j) + ".bin", wire::serialize(fo.speed));
            if (fo.path_updated) spit(outdir + "/path_" + std::to_string(idx) + "_" + std::to_string(j) + ".bin", wire::serialize(TF.path));
          }
          std::printf("fusion %d %d %d\n", idx, j, pub ? 1 : 0);
          j++;
        }
        ros1::TransformFusionNode::PredictOutputs po;
        if (TF.predictTimerHandler(stamp, po)) spit(outdir + "/future_" + std::to_string(idx) + ".bin", wire::serialize(po.future_pose_lidar));
      };
      int processed = 0;
      std::vector<wire::Time> seen_stamps;
      for (int k = 8; k < argc; k++) {
        const int idx = k - 8;
        const std::vector<uint8_t> raw = slurp(argv[k]);
        wire::PointCloud2 msg;
        if (!wire::deserialize(raw.data(), raw.size(), msg)) { std::printf("msg %d malformed\n", idx); return 3; }
        seen_stamps.push_back(msg.header.stamp);
        if (idx == backend_at) {   // the back end's pose belongs to an OLDER scan (three messages back), as its latency makes it in the reference
          wire::Odometry mapped; mapped.header.stamp = seen_stamps[idx >= 3 ? idx - 3 : 0]; mapped.pose.orientation[3] = 1.0;
          C.odometryHandler(mapped); TF.mappingOdometryHandler(mapped);
        }
        wire::CloudInfoStamp infoA;
        const ros1::Status sa = A.cloudHandler(msg, infoA);
        if (sa != ros1::Status::Published) { std::printf("msg %d imageProjection %d\n", idx, (int)sa); tick(idx, msg.header.stamp); continue; }
        const std::vector<uint8_t> hopAB = wire::serialize(infoA);
        spit(outdir + "/cloud_info_" + std::to_string(processed) + ".bin", hopAB);
        wire::CloudInfoStamp inB, infoB;
        if (!wire::deserialize(hopAB.data(), hopAB.size(), inB)) return 3;
        const ros1::Status sb = B.laserCloudInfoHandler(inB, infoB);
        if (sb != ros1::Status::Published) { std::printf("msg %d featureExtraction %d\n", idx, (int)sb); continue; }
        const std::vector<uint8_t> hopBC = wire::serialize(infoB);
        spit(outdir + "/feature_info_" + std::to_string(processed) + ".bin", hopBC);
        wire::CloudInfoStamp inC;
        if (!wire::deserialize(hopBC.data(), hopBC.size(), inC)) return 3;
        ros1::LidarOdometryNode::Outputs o;
        const ros1::Status sc = C.cloudHandler(inC, inC.header.stamp /* replay: "now" = the stamp of the cloud */, o);
        if (sc == ros1::Status::Published) {
          spit(outdir + "/odom_" + std::to_string(processed) + ".bin", wire::serialize(o.laser_odom_incremental));
          spit(outdir + "/pose_" + std::to_string(processed) + ".bin", wire::serialize(o.laser_pose));
          spit(outdir + "/odom_cloud_" + std::to_string(processed) + ".bin", wire::serialize(o.odometry_cloud));
          TF.lidarOdometryHandler(o.laser_odom_incremental);   // odomTopic + "_incremental" loops back into the same process
          A.odometryHandler(o.laser_odom_incremental);         // ... and into ImageProjection (imageProjection.cpp:92, :149-155): the de-skew's odometry
        }
        std::printf("msg %d cloud %d lidarOdometry %d frame %d\n", idx, processed, (int)sc, (int)o.frame);
        processed++;
        tick(idx, msg.header.stamp);
      }
      // error paths of cachePointCloud: a non-dense cloud and a cloud without a ring field
      {
        const std::vector<uint8_t> raw = slurp(argv[8]);
        wire::PointCloud2 msg; wire::deserialize(raw.data(), raw.size(), msg);
        wire::CloudInfoStamp tmp;
        Context ctxD; ros1::ImageProjectionNode D(ctxD, P);
        wire::PointCloud2 bad = msg; bad.is_dense = 0;
        D.cloudHandler(bad, tmp); D.cloudHandler(bad, tmp);
        std::printf("nondense %d\n", (int)D.cloudHandler(bad, tmp));
        Context ctxE; ros1::ImageProjectionNode E(ctxE, P);
        wire::PointCloud2 noring = msg;
        for (auto& f : noring.fields) if (f.name == "ring") f.name = "rng";
        E.cloudHandler(noring, tmp); E.cloudHandler(noring, tmp);
        std::printf("noring %d\n", (int)E.cloudHandler(noring, tmp));
      }
    } catch (const rolo::Error& e) {
      std::fprintf(stderr, "rolo::Error %d: %s\n", e.code, e.what());
      return 4;
    }
    return 0;
  }
  return 1;
}
