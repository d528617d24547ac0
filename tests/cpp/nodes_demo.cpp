// C++ host-side check of the node-core mirrors (include/rolo_nodes_hip.hpp): a raw frame sequence goes through
// (a) ImageProjection -> FeatureExtraction -> LidarOdometry::cloudHandler, the way the three ROS nodes chain them, and
// (b) LidarOdometry::submit / collect, the fused device-resident path — in a C++-only process (no Python / torch).
// Input: frames.bin = int32 n_frames, then per frame int32 n, n x 3 float32 xyz, n x uint16 ring. Prints one line per
// frame and path: status, LaserOdomPose[6], Translation[3], n_corner, n_surface.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rolo_nodes_hip.hpp"

struct Frame { std::vector<float> xyz; std::vector<uint16_t> ring; int n; };

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: nodes_demo frames.bin N_SCAN Horizon_SCAN\n"); return 1; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  int32_t nf = 0;
  if (std::fread(&nf, 4, 1, f) != 1) return 3;
  std::vector<Frame> frames(nf);
  for (auto& fr : frames) {
    int32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return 3;
    fr.n = n; fr.xyz.resize((size_t)n * 3); fr.ring.resize(n);
    if (std::fread(fr.xyz.data(), 12, n, f) != (size_t)n || std::fread(fr.ring.data(), 2, n, f) != (size_t)n) return 3;
  }
  std::fclose(f);
  const rolo::FrontParams fp(std::atoi(argv[2]), std::atoi(argv[3]));
  try {
    rolo::Context staged_ctx, fused_ctx;
    rolo::ImageProjection projection(staged_ctx, fp);
    rolo::FeatureExtraction features(staged_ctx, fp);
    rolo::LidarOdometry staged(staged_ctx, 0.3f), fused(fused_ctx, 0.3f);
    fused.submit(fp, 100.0, frames[0].xyz.data(), 3, frames[0].ring.data(), frames[0].n);
    for (int k = 0; k < nf; k++) {
      const double stamp = 100.0 + 0.1 * k;
      if (k == 2) { staged.odometryHandler(stamp - 0.05); fused.odometryHandler(stamp - 0.05); }  // the back end's first odometry
      const rolo::CloudInfo& info = projection.projectPointCloud(frames[k].xyz.data(), 3, frames[k].ring.data(), frames[k].n);
      features.extractFeatures(info.n_valid);
      const int st = staged.cloudHandler(stamp, features.cornerCloud, features.surfaceCloud);
      std::printf("staged %d", st);
      for (float v : staged.LaserOdomPose) std::printf(" %.9g", v);
      for (double v : staged.Translation) std::printf(" %.17g", v);
      std::printf(" %zu %zu\n", features.cornerCloud.size() / 4, features.surfaceCloud.size() / 4);
      if (k + 1 < nf) fused.submit(fp, stamp + 0.1, frames[k + 1].xyz.data(), 3, frames[k + 1].ring.data(), frames[k + 1].n);
      const int sf = fused.collect();
      std::printf("fused %d", sf);
      for (float v : fused.LaserOdomPose) std::printf(" %.9g", v);
      for (double v : fused.Translation) std::printf(" %.17g", v);
      std::printf(" %d %d\n", fused.counts[1], fused.counts[2]);
    }
    // error behaviour: misuse surfaces as rolo::Error with the ROLO_E* code
    try { fused.collect(); std::printf("no-throw\n"); } catch (const rolo::Error& e) { std::printf("error %d\n", e.code); }
  } catch (const rolo::Error& e) {
    std::fprintf(stderr, "rolo::Error %d: %s\n", e.code, e.what());
    return 4;
  }
  return 0;
}
