// dump_reference_golden — pins the oracle from OUTSIDE this repository.
//
// This program is built against the REAL reference: sdwyc/ROLO's rot_gicp headers + sources, PCL (+ FLANN), Eigen — in a ROLO catkin
// workspace (or any machine that has them). It CANNOT be built or run in this repository's image (no Eigen, PCL or ROS here; see
// DESIGN.md "Oracle"), so it is not exercised by the test suite; what the suite does hold is the other half of the job:
// tests/test_oracle_vs_reference_dump.py compares the oracle with the files this program writes as soon as tests/golden/ref_<case>.npz
// exist. Recipe: tools/README.md.
//
// For one fixture case it reads the inputs exported by tools/export_golden_inputs.py (raw little-endian arrays + params.txt), runs the
// reference's own fast_gicp::RotVGICP<pcl::PointXYZI, pcl::PointXYZI> on them through a subclass that makes the protected stage
// functions callable, and writes every per-stage field tests/golden/make_golden.py stores (plus full-length ones where the twin only
// sampled) as .npy files:
//   src_cov, tgt_cov            calculate_covariances            rot_vgicp_impl.hpp:421-496
//   tgt_keys, vox_*             create_voxelmap / lookup_voxel   vmp_voxel.hpp:167-233
//   so3_err/H/b, corr_*         so3_linearize + update_correspondences   :173-222, :293-388
//   err_probe2                  compute_error                    :391-417
//   lin6_err/H/b                linearize                        :225-290
//   t3_err/H/b, t3_err_variant  t3_linearize / compute_t_error   :499-658
//   align_T(_f), align_iters, align_converged     pcl::Registration::align -> computeTransformation (lsq_registration_impl.hpp:152-179)
//   trans_final                 computeTranslation               :55-80
//
//   g++ -O3 -DNDEBUG -std=c++14 -fopenmp -I<ROLO>/include $(pkg-config --cflags eigen3 pcl_registration-1.10) [line continues]
//       tools/dump_reference_golden.cpp <ROLO>/src/rot_gicp/gicp/*.cpp -o dump_reference_golden $(pkg-config --libs pcl_registration-1.10 pcl_search-1.10 pcl_kdtree-1.10) -lflann_cpp
//   ./dump_reference_golden <inputs_dir> <outputs_dir>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

// -DDUMP_IO_SELFTEST: only the half that needs nothing of the reference — the raw-array reader, the .npy writer, the parameter file — with a main that
// round-trips them (tests/test_oracle_vs_reference_dump.py builds and runs it in this repository's image, numpy reads the files back: a typo or a broken
// .npy header there would otherwise greet the first maintainer who runs the recipe). The other half cannot be compiled without Eigen / PCL.
#ifndef DUMP_IO_SELFTEST
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <rot_gicp/gicp/rot_vgicp.hpp>
#include <rot_gicp/gicp/impl/lsq_registration_impl.hpp>
#include <rot_gicp/gicp/impl/rot_vgicp_impl.hpp>

using PointT = pcl::PointXYZI;
using Base = fast_gicp::RotVGICP<PointT, PointT>;
#endif

// ---- minimal I/O: raw arrays in, .npy (format 1.0) out ----------------------------------------------------------------------------------
template <typename T> static std::vector<T> read_raw(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
  const std::streamsize bytes = f.tellg(); f.seekg(0);
  std::vector<T> v((size_t)bytes / sizeof(T));
  f.read(reinterpret_cast<char*>(v.data()), bytes);
  return v;
}
template <typename T> struct NpyType;
template <> struct NpyType<double> { static const char* descr() { return "<f8"; } };
template <> struct NpyType<float> { static const char* descr() { return "<f4"; } };
template <> struct NpyType<int32_t> { static const char* descr() { return "<i4"; } };
template <typename T> static void write_npy(const std::string& path, const std::vector<T>& v, const std::vector<size_t>& shape) {
  std::ostringstream h;
  h << "{'descr': '" << NpyType<T>::descr() << "', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < shape.size(); i++) h << shape[i] << (shape.size() == 1 ? "," : (i + 1 < shape.size() ? ", " : ""));
  h << "), }";
  std::string hdr = h.str();
  while ((10 + hdr.size() + 1) % 64 != 0) hdr += ' ';
  hdr += '\n';
  std::ofstream f(path, std::ios::binary);
  const char magic[8] = {'\x93', 'N', 'U', 'M', 'P', 'Y', 1, 0};
  f.write(magic, 8);
  const uint16_t len = (uint16_t)hdr.size();
  f.write(reinterpret_cast<const char*>(&len), 2);
  f.write(hdr.data(), (std::streamsize)hdr.size());
  f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}
static std::map<std::string, double> read_params(const std::string& path) {
  std::map<std::string, double> m; std::ifstream f(path); std::string k; double v;
  while (f >> k >> v) m[k] = v;
  return m;
}
#ifdef DUMP_IO_SELFTEST
int main(int argc, char** argv) {
  if (argc != 2) { std::fprintf(stderr, "usage: dump_io_selftest <dir>\n"); return 1; }
  const std::string d = std::string(argv[1]) + "/";
  auto P = read_params(d + "params.txt");
  const std::vector<float> a = read_raw<float>(d + "source.f32");
  const std::vector<double> t = read_raw<double>(d + "T_probe.f64");
  write_npy(d + "echo_source.npy", a, {a.size() / 4, 4});
  write_npy(d + "echo_T.npy", t, {4, 4});
  write_npy(d + "echo_scalar.npy", std::vector<double>{P["ct_lambda"] + P["leaf"]}, {1});
  write_npy(d + "echo_ints.npy", std::vector<int32_t>{(int32_t)P["voxel_type"], 7, -3}, {3});
  std::printf("ok %zu %zu\n", a.size(), t.size());
  return 0;
}
#else
static pcl::PointCloud<PointT>::Ptr to_cloud(const std::vector<float>& a) {
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>);
  c->resize(a.size() / 4);
  for (size_t i = 0; i < a.size() / 4; i++) { PointT& p = c->points[i]; p.x = a[4 * i]; p.y = a[4 * i + 1]; p.z = a[4 * i + 2]; p.intensity = a[4 * i + 3]; }
  return c;
}
static Eigen::Isometry3d iso(const std::vector<double>& T16) {
  Eigen::Isometry3d x = Eigen::Isometry3d::Identity();
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) x.matrix()(i, j) = T16[i * 4 + j];
  return x;
}
template <typename M> static std::vector<double> flat(const M& m) {
  std::vector<double> v; for (int i = 0; i < m.rows(); i++) for (int j = 0; j < m.cols(); j++) v.push_back(m(i, j)); return v;
}

// ---- the reference operator with its protected stages opened up ---------------------------------------------------------------------------
struct Dump : public Base {
  using Base::so3_linearize; using Base::linearize; using Base::compute_error; using Base::t3_linearize; using Base::compute_t_error;
  using Base::voxelmap_; using Base::voxel_correspondences_; using Base::voxel_type_; using Base::input_; using Base::target_;
  using Base::nr_iterations_; using Base::final_transformation_;
};

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: dump_reference_golden <inputs_dir> <outputs_dir>\n"); return 1; }
  const std::string in = std::string(argv[1]) + "/", out = std::string(argv[2]) + "/";
  auto P = read_params(in + "params.txt");
  auto src = to_cloud(read_raw<float>(in + "source.f32")), tgt = to_cloud(read_raw<float>(in + "target.f32"));
  const size_t ns = src->size(), nt = tgt->size();
  auto make = [&]() {
    std::unique_ptr<Dump> g(new Dump);
    if ((int)P["voxel_type"] == 0) g->setPolarResolution(P["polar_theta"], P["polar_phi"], P["polar_r"]); else g->setResolution(P["leaf"]);
    g->setNumThreads(1);   // one thread: sums in point order, as the oracle's single-thread mode
    g->clearTarget(); g->clearSource();
    g->setInputTarget(tgt); g->setInputSource(src);
    return g;
  };
  // ---- stage level, on one operator --------------------------------------------------------------------------------------------------------
  auto g = make();
  const Eigen::Isometry3d Tp = iso(read_raw<double>(in + "T_probe.f64")), Tp2 = iso(read_raw<double>(in + "T_probe2.f64")), Tp6 = iso(read_raw<double>(in + "T_probe6.f64"));
  Eigen::Matrix3d H3; Eigen::Vector3d b3;
  const double e3 = g->so3_linearize(Tp, &H3, &b3);   // computes covariances + voxel map + correspondences at T_probe on first use
  {
    std::vector<double> c; for (const auto& m : g->getSourceCovariances()) { auto f = flat(m); c.insert(c.end(), f.begin(), f.end()); }
    write_npy(out + "src_cov.npy", c, {ns, 4, 4});
    c.clear(); for (const auto& m : g->getTargetCovariances()) { auto f = flat(m); c.insert(c.end(), f.begin(), f.end()); }
    write_npy(out + "tgt_cov.npy", c, {nt, 4, 4});
  }
  {   // voxel of every target point: key, and the finalized voxel it landed in (count, mean, cov)
    std::vector<int32_t> keys; std::vector<double> mean, cov; std::vector<int32_t> cnt;
    for (size_t i = 0; i < nt; i++) {
      const Eigen::Vector4d x = tgt->at(i).getVector4fMap().cast<double>();
      const Eigen::Vector3i k = g->voxel_type_ == fast_gicp::VoxelType::POLAR ? g->voxelmap_->polar_coord(x) : g->voxelmap_->voxel_coord(x);
      keys.push_back(k[0]); keys.push_back(k[1]); keys.push_back(k[2]);
      const auto v = g->voxelmap_->lookup_voxel(k);
      cnt.push_back(v ? v->num_points : 0);
      for (int d = 0; d < 4; d++) mean.push_back(v ? v->mean_dir[d] : 0.0);
      for (int r = 0; r < 4; r++) for (int c2 = 0; c2 < 4; c2++) cov.push_back(v ? v->cov(r, c2) : 0.0);
    }
    write_npy(out + "tgt_keys.npy", keys, {nt, 3}); write_npy(out + "tgt_vox_count.npy", cnt, {nt});
    write_npy(out + "tgt_vox_mean.npy", mean, {nt, 4}); write_npy(out + "tgt_vox_cov.npy", cov, {nt, 4, 4});
  }
  write_npy(out + "so3_err.npy", std::vector<double>{e3}, {1}); write_npy(out + "so3_H.npy", flat(H3), {3, 3}); write_npy(out + "so3_b.npy", flat(b3), {3});
  {   // correspondences of that linearisation: source index + the key of the voxel (recomputed from the transformed point, as update_correspondences does)
    std::vector<int32_t> cs, ck;
    for (const auto& c : g->voxel_correspondences_) {
      cs.push_back(c.first);
      const Eigen::Vector4d x = Tp * src->at(c.first).getVector4fMap().cast<double>();
      const Eigen::Vector3i k = g->voxel_type_ == fast_gicp::VoxelType::POLAR ? g->voxelmap_->polar_coord(x) : g->voxelmap_->voxel_coord(x);
      ck.push_back(k[0]); ck.push_back(k[1]); ck.push_back(k[2]);
    }
    write_npy(out + "corr_src.npy", cs, {cs.size()}); write_npy(out + "corr_vox_keys.npy", ck, {cs.size(), 3});
  }
  write_npy(out + "err_probe2.npy", std::vector<double>{g->compute_error(Tp2)}, {1});
  {   // translation-stage pieces on the correspondences cached by that so3_linearize
    const auto tg = read_raw<double>(in + "t_guess.f64"), tl = read_raw<double>(in + "t_last.f64"), tp = read_raw<double>(in + "t_probe.f64");
    const Eigen::Vector3d G(tg[0], tg[1], tg[2]), L(tl[0], tl[1], tl[2]), T(tp[0], tp[1], tp[2]);
    Eigen::Matrix<double, 6, 6> H6; Eigen::Matrix<double, 6, 1> b6;
    // lambda_ (the CT weight) is set by computeTranslation only; reach it the same way the oracle's fixtures do: ct_lambda from params
    {
      // (t3_linearize reads lambda_; set it without running a solve: the member is protected, this subclass may write it)
      struct L_ : Dump { static void set(Dump& d, float v) { static_cast<L_&>(d).lambda_ = v; } };
      L_::set(*g, (float)P["ct_lambda"]);
    }
    const double et = g->t3_linearize(T, G, L, 0.1, 0.1, &H6, &b6);
    const double ev = g->compute_t_error(T, G, L, 0.1, 0.1);
    write_npy(out + "t3_err.npy", std::vector<double>{et}, {1}); write_npy(out + "t3_H.npy", flat(H6), {6, 6}); write_npy(out + "t3_b.npy", flat(b6), {6});
    write_npy(out + "t3_err_variant.npy", std::vector<double>{ev}, {1});
  }
  {
    Eigen::Matrix<double, 6, 6> H6; Eigen::Matrix<double, 6, 1> b6;
    const double e6 = g->linearize(Tp6, &H6, &b6);
    write_npy(out + "lin6_err.npy", std::vector<double>{e6}, {1}); write_npy(out + "lin6_H.npy", flat(H6), {6, 6}); write_npy(out + "lin6_b.npy", flat(b6), {6});
  }
  // ---- the full solve, on a fresh operator exactly as scanRegeistration drives it (lidarOdometry.cpp:460-494) ---------------------------------
  {
    auto h = make();
    pcl::PointCloud<PointT> aligned;
    h->align(aligned);
    const Eigen::Matrix4f Tf = h->getFinalTransformation();
    write_npy(out + "align_T_f.npy", std::vector<float>(Tf.data(), Tf.data() + 16), {4, 4});   // column-major in memory: the packer transposes
    write_npy(out + "align_iters.npy", std::vector<int32_t>{h->nr_iterations_ + 1, h->hasConverged() ? 1 : 0}, {2});
    const auto tg = read_raw<double>(in + "t_guess.f64"), tl = read_raw<double>(in + "t_last.f64");
    Eigen::Vector3d reg = Eigen::Vector3d::Zero();
    h->computeTranslation(aligned, reg, Eigen::Vector3d(tg[0], tg[1], tg[2]), Eigen::Vector3d(tl[0], tl[1], tl[2]), 0.1, 0.1, (float)P["ct_lambda"]);
    write_npy(out + "trans_final.npy", flat(reg), {3});
  }
  std::printf("wrote the reference's per-stage fields for %zu + %zu points to %s\n", ns, nt, out.c_str());
  return 0;
}
#endif   // DUMP_IO_SELFTEST
