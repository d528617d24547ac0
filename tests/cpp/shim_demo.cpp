// C++ host-side check of the drop-in: drives fast_gicp::RotVGICP (include/rot_vgicp_hip.hpp) exactly as
// LidarOdometry::scanRegeistration does (reference src/lidarOdometry.cpp:460-500), with no Python / torch in the process.
// Reads two clouds (n x 4 float32: x y z intensity) from binary files, prints the results as one line of numbers.
//
//   shim_demo source.bin target.bin              one registration + the accessor / error-behaviour checks (three lines of output)
//   shim_demo loop source.bin target.bin N       N frames three ways, milliseconds per frame: the operator constructed INSIDE the frame as
//                                                lidarOdometry.cpp:460 does (the library's context pool behind the constructor), one persistent
//                                                operator, and a raw rolo_ctx_create / rolo_ctx_destroy per frame (no pool: what round 2 shipped)
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
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

// scanRegeistration's use of the operator (src/lidarOdometry.cpp:460-494) on a given object
static void one_frame(fast_gicp::RotVGICP<>& rot_vgicp, const rolo::Cloud::Ptr& source, const rolo::Cloud::Ptr& target, std::array<float, 16>& T, std::array<double, 3>& reg_t) {
  rolo::Cloud aligned;
  rot_vgicp.setPolarResolution(0.175, 0.175, 2.0);
  rot_vgicp.setNumThreads(0);
  rot_vgicp.clearTarget(); rot_vgicp.clearSource();
  rot_vgicp.setInputTarget(target);
  rot_vgicp.setInputSource(source);
  rot_vgicp.align(aligned);
  T = rot_vgicp.getFinalTransformation();
  reg_t = {0, 0, 0};
  const std::array<double, 3> guess{-0.28, -0.04, -0.02}, last{-0.28, -0.04, -0.02};
  rot_vgicp.computeTranslation(aligned, reg_t, guess, last, 0.1, 0.1, 0.3f);
}

static int loop_mode(const char* srcp, const char* tgtp, int n) {
  rolo::Cloud::Ptr source = load(srcp), target = load(tgtp);
  using clk = std::chrono::steady_clock;
  std::array<float, 16> T0{}, T{}; std::array<double, 3> t0{}, t{};
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  // warm-up: allocations, code objects
  for (int k = 0; k < 3; k++) { fast_gicp::RotVGICP<> g; one_frame(g, source, target, T0, t0); }
  // (a) the operator constructed inside the frame, as the reference does: context pool behind the constructor
  auto a0 = clk::now();
  for (int k = 0; k < n; k++) { fast_gicp::RotVGICP<> g; one_frame(g, source, target, T, t); }
  auto a1 = clk::now();
  const bool same_a = std::memcmp(T.data(), T0.data(), sizeof(float) * 16) == 0 && t == t0;
  // (b) one persistent operator
  double pb;
  { fast_gicp::RotVGICP<> g; one_frame(g, source, target, T, t);
    auto b0 = clk::now();
    for (int k = 0; k < n; k++) one_frame(g, source, target, T, t);
    pb = ms(b0, clk::now()) / n; }
  const bool same_b = std::memcmp(T.data(), T0.data(), sizeof(float) * 16) == 0 && t == t0;
  // (c) what an unpooled constructor costs on top: a context created and destroyed (streams, events, pinned + device buffers)
  rolo_ctx_pool_clear();
  auto c0 = clk::now();
  const int nc = n < 10 ? n : 10;
  for (int k = 0; k < nc; k++) { rolo_ctx* c = nullptr; if (rolo_ctx_create(0, &c) != ROLO_OK) return 5; rolo_ctx_destroy(c); }
  const double pc = ms(c0, clk::now()) / nc;
  std::printf("%.4f %.4f %.4f %d %d\n", ms(a0, a1) / n, pb, pc, same_a ? 1 : 0, same_b ? 1 : 0);
  return 0;
}

// setDebugPrint(true): the reference's per-trial LM tables on stdout (lsq_registration_impl.hpp:158-162, :299-305, :114-120)
static int debug_mode(const char* src_path, const char* tgt_path) {
  rolo::Cloud::Ptr source = load(src_path), target = load(tgt_path);
  rolo::Cloud aligned;
  fast_gicp::RotVGICP<> rot_vgicp;
  rot_vgicp.setPolarResolution(0.175, 0.175, 2.0);
  rot_vgicp.setDebugPrint(true);
  rot_vgicp.setInputTarget(target); rot_vgicp.setInputSource(source);
  rot_vgicp.align(aligned);
  std::printf("=== translation ===\n");
  std::array<double, 3> reg_t{0, 0, 0}, guess{-0.28, -0.04, -0.02}, last{-0.28, -0.04, -0.02};
  rot_vgicp.computeTranslation(aligned, reg_t, guess, last, 0.1, 0.1, 0.3f);
  return 0;
}

// a stage that gives up: "lm not converged!!" on stderr (lsq_registration_impl.hpp:66-69, :166-169), the result still returned. lm_max_iterations_ is protected
// in the reference (lsq_registration.hpp:104): a subclass sets it, there as here
struct FewTrials : fast_gicp::RotVGICP<> { void setTrials(int n) { set_lm_max_iterations(n); } };
static int lmfail_mode(const char* src_path, const char* tgt_path, int trials) {
  rolo::Cloud::Ptr source = load(src_path), target = load(tgt_path);
  rolo::Cloud aligned;
  FewTrials rot_vgicp;
  rot_vgicp.setPolarResolution(0.175, 0.175, 2.0);
  rot_vgicp.setTrials(trials);
  rot_vgicp.setInputTarget(target); rot_vgicp.setInputSource(source);
  rot_vgicp.align(aligned);
  std::array<double, 3> reg_t{0, 0, 0}, guess{-0.28, -0.04, -0.02}, last{-0.28, -0.04, -0.02};
  rot_vgicp.computeTranslation(aligned, reg_t, guess, last, 0.1, 0.1, 0.3f);
  std::printf("%.17g %.17g %.17g %d %zu\n", reg_t[0], reg_t[1], reg_t[2], rot_vgicp.hasConverged() ? 1 : 0, aligned.size());
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 5 && std::strcmp(argv[1], "lmfail") == 0) return lmfail_mode(argv[2], argv[3], std::atoi(argv[4]));
  if (argc == 5 && std::strcmp(argv[1], "loop") == 0) return loop_mode(argv[2], argv[3], std::atoi(argv[4]));
  if (argc == 4 && std::strcmp(argv[1], "debug") == 0) return debug_mode(argv[2], argv[3]);
  if (argc < 3) { std::fprintf(stderr, "usage: shim_demo source.bin target.bin | shim_demo loop source.bin target.bin N\n"); return 1; }
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
  // `aligned` after computeTranslation = pcl::transformPointCloud(*input_, output, translation) (lsq_registration_impl.hpp:75-78): the class computes it on the host;
  // rolo_transform_cloud is the same restatement on the device (bit-exact against the oracle's) — the two must agree bit for bit, all 8 floats of every record
  int aligned_same = 0;
  {
    std::vector<rolo::PointXYZI> ref(source->points.size());
    const float Tt[16] = {1, 0, 0, (float)reg_t[0], 0, 1, 0, (float)reg_t[1], 0, 0, 1, (float)reg_t[2], 0, 0, 0, 1};
    if (rolo_transform_cloud(rot_vgicp.handle(), reinterpret_cast<const float*>(source->points.data()), reinterpret_cast<float*>(ref.data()), (int)ref.size(), 8, Tt) == ROLO_OK)
      aligned_same = aligned.points.size() == ref.size() && std::memcmp(aligned.points.data(), ref.data(), sizeof(rolo::PointXYZI) * ref.size()) == 0;
  }
  std::printf("%.17g %.17g %.17g %d %zu %d\n", reg_t[0], reg_t[1], reg_t[2], rot_vgicp.hasConverged() ? 1 : 0, aligned.size(), aligned_same);
  // covariance accessors, getFinalHessian, evaluateCost (rot_vgicp.hpp:89-97, lsq_registration.hpp:55-57): a second operator fed with the
  // first one's covariances must reproduce its rotation; line 3 = max |dT|, cost at identity, trace(H), |b|, final Hessian (0,0), n_covs
  {
    const auto& cs = rot_vgicp.getSourceCovariances();
    const auto& ct = rot_vgicp.getTargetCovariances();
    fast_gicp::RotVGICP<> second;
    second.setPolarResolution(0.175, 0.175, 2.0);
    second.setInputTarget(target); second.setInputSource(source);
    second.setSourceCovariances(cs); second.setTargetCovariances(ct);
    rolo::Cloud aligned2;
    second.align(aligned2);
    auto T2 = second.getFinalTransformation();
    double dmax = 0; for (int i = 0; i < 16; i++) { const double d = std::fabs((double)T2[i] - (double)T[i]); if (d > dmax) dmax = d; }
    fast_gicp::RotVGICP<>::Matrix4 I{}; for (int i = 0; i < 16; i++) I[i] = (i % 5 == 0) ? 1.f : 0.f;
    fast_gicp::RotVGICP<>::Matrix6d H; fast_gicp::RotVGICP<>::Vector6d b;
    const double cost = second.evaluateCost(I, &H, &b);
    double tr = 0, bn = 0; for (int i = 0; i < 6; i++) { tr += H[i * 6 + i]; bn += b[i] * b[i]; }
    std::printf("%.3g %.17g %.17g %.17g %.17g %zu\n", dmax, cost, tr, std::sqrt(bn), second.getFinalHessian()[0], cs.size());
  }
  // error behaviour of the reference: aliasing the output with an input throws std::invalid_argument
  try { rot_vgicp.align(const_cast<rolo::Cloud&>(*source)); std::printf("no-throw\n"); }
  catch (const std::invalid_argument&) { std::printf("invalid_argument\n"); }
  return 0;
}
