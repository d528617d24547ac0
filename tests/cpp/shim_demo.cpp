// C++ host-side check of the drop-in: drives fast_gicp::RotVGICP (include/rot_vgicp_hip.hpp) exactly as
// LidarOdometry::scanRegeistration does (reference src/lidarOdometry.cpp:460-500), with no Python / torch in the process.
// Reads two clouds (n x 4 float32: x y z intensity) from binary files, prints the results as one line of numbers.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rot_vgicp_hip.hpp"

static rolo::Cloud::Ptr load(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  std::fseek(f, 0, SEEK_END); long bytes = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  const size_t n = (size_t)bytes / 16;
  std::vector<float> raw(n * 4);
  if (std::fread(raw.data(), 16, n, f) != n) std::exit(3);
  std::fclose(f);
  auto c = std::make_shared<rolo::Cloud>();
  c->points.resize(n);
  for (size_t i = 0; i < n; i++) { auto& p = c->points[i]; p.x = raw[4 * i]; p.y = raw[4 * i + 1]; p.z = raw[4 * i + 2]; p.w = 1.f; p.intensity = raw[4 * i + 3]; p.pad[0] = p.pad[1] = p.pad[2] = 0; }
  return c;
}

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: shim_demo source.bin target.bin\n"); return 1; }
  rolo::Cloud::Ptr source = load(argv[1]), target = load(argv[2]);
  rolo::Cloud aligned;
  fast_gicp::RotVGICP<> rot_vgicp;
  rot_vgicp.setPolarResolution(0.175, 0.175, 2.0);
  rot_vgicp.setNumThreads(0);
  rot_vgicp.clearTarget(); rot_vgicp.clearSource();
  rot_vgicp.setInputTarget(target);
  rot_vgicp.setInputSource(source);
  rot_vgicp.align(aligned);
  auto T = rot_vgicp.getFinalTransformation();
  std::array<double, 3> reg_t{0, 0, 0}, guess{-0.28, -0.04, -0.02}, last{-0.28, -0.04, -0.02};
  rot_vgicp.computeTranslation(aligned, reg_t, guess, last, 0.1, 0.1, 0.3f);
  for (int i = 0; i < 16; i++) std::printf("%.9g ", T[i]);
  std::printf("%.17g %.17g %.17g %d %zu\n", reg_t[0], reg_t[1], reg_t[2], rot_vgicp.hasConverged() ? 1 : 0, aligned.size());
  // error behaviour of the reference: aliasing the output with an input throws std::invalid_argument
  try { rot_vgicp.align(const_cast<rolo::Cloud&>(*source)); std::printf("no-throw\n"); }
  catch (const std::invalid_argument&) { std::printf("invalid_argument\n"); }
  return 0;
}
