// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product (see orc_linalg.hpp header).
// PARITY UNPINNED by the reference itself.
//
// CPU restatement of: src/imageProjection.cpp:399-505 (K1 projectPointCloud, K2 cloudExtraction),
// src/featureExtraction.cpp:87-266 (K3 calculateSmoothness + markOccludedPoints, K4 extractFeatures incl.
// pcl::VoxelGrid), src/lidarOdometry.cpp:448-626,700-712 (K13 per-frame driver). PCL helpers
// (VoxelGrid, getTransformation, getTranslationAndEulerAngles, transformPointCloud) are restated from
// their documented behaviour (PCL >=1.10 is not under /root/reference).
#include "rolo_oracle_front.h"
#include "orc_linalg.hpp"
#include <cmath>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>

extern "C" {

void orc_front_default_params(orc_front_params* p) {  // config/params.yaml:20-36
  p->n_scan = 32; p->horizon_scan = 1024; p->downsample_rate = 1;
  p->lidar_min_range = 2.0f; p->lidar_max_range = 1000.0f;
  p->edge_threshold = 0.8f; p->surf_threshold = 0.1f; p->odometry_surf_leaf_size = 0.4f;
}

// imageProjection.cpp:399-460 + :477-505
// ImageProjection::deskewPoint :368-396, rotation only
static void deskew_point(const orc_deskew* d, float rel_time_f, float& x, float& y, float& z) {
  const double relTime = (double)rel_time_f;           // the float stored in deskewCloud->points[i].intensity, passed as double
  const float ratio = relTime / d->scan_period;        // :380 double / float, stored float
  const float s1 = (float)(d->scan_period / d->odom_time_diff);  // :383 float / double; Eigen casts the scalar to the vector's float
  const float r = d->odom_incre_rpy[0] * s1 * ratio, p = d->odom_incre_rpy[1] * s1 * ratio, w = d->odom_incre_rpy[2] * s1 * ratio;
  float T[16];
  orc_get_transformation(0.f, 0.f, 0.f, -r, -p, -w, T);  // :386 pcl::getTransformation (float overload)
  const float nx = T[0] * x + T[1] * y + T[2] * z + T[3];  // :389-391
  const float ny = T[4] * x + T[5] * y + T[6] * z + T[7];
  const float nz = T[8] * x + T[9] * y + T[10] * z + T[11];
  x = nx; y = ny; z = nz;
}

void orc_azimuth_times(const float* pts, int stride, int n, float scan_period, float* rel_time) {  // imageProjection.cpp:270-327
  if (n <= 0) return;
  bool halfPassed = false;
  float startOri = -atan2f(pts[1], pts[0]);                                                       // :272
  float endOri = -atan2f(pts[(size_t)(n - 1) * stride + 1], pts[(size_t)(n - 1) * stride]) + 2 * M_PI;   // :273 (double sum, stored float)
  if (endOri - startOri > 3 * M_PI) endOri -= 2 * M_PI;                                           // :274-277
  else if (endOri - startOri < M_PI) endOri += 2 * M_PI;
  const float orientationDiff = endOri - startOri;
  for (int i = 0; i < n; i++) {
    const float px = pts[(size_t)i * stride + 1], pz = pts[(size_t)i * stride];                   // point.x = in.y, point.z = in.x (:303-305)
    float ori = -atan2f(px, pz);                                                                   // :307
    if (!halfPassed) {
      if (ori < startOri - M_PI / 2) ori += 2 * M_PI;
      else if (ori > startOri + M_PI * 3 / 2) ori -= 2 * M_PI;
      if (ori - startOri > M_PI) halfPassed = true;
    } else {
      ori += 2 * M_PI;
      if (ori < endOri - M_PI * 3 / 2) ori += 2 * M_PI;
      else if (ori > endOri + M_PI / 2) ori -= 2 * M_PI;
    }
    const float relTime = (ori - startOri) / orientationDiff;                                     // :324
    rel_time[i] = scan_period * relTime;                                                           // :325
  }
}

int orc_project(const orc_front_params* P, const float* pts, int stride, const uint16_t* ring, int n_raw,
                float* range_mat, float* full_cloud, float* extracted, int32_t* point_col_ind,
                float* point_range, int32_t* start_ring, int32_t* end_ring) {
  return orc_project_deskew(P, pts, stride, ring, n_raw, nullptr, nullptr, range_mat, full_cloud, extracted, point_col_ind, point_range, start_ring, end_ring);
}

int orc_project_deskew(const orc_front_params* P, const float* pts, int stride, const uint16_t* ring, int n_raw,
                       const float* rel_time, const orc_deskew* d,
                       float* range_mat, float* full_cloud, float* extracted, int32_t* point_col_ind,
                       float* point_range, int32_t* start_ring, int32_t* end_ring) {
  const int NS = P->n_scan, H = P->horizon_scan;
  if (NS <= 0 || H <= 0 || P->downsample_rate <= 0) return -1;
  for (int i = 0; i < NS * H; i++) range_mat[i] = FLT_MAX;  // :130
  memset(full_cloud, 0, sizeof(float) * 4 * (size_t)NS * H);
  const float ang_res_x = 360.0 / float(H);  // :438 (double division, stored float)
  for (int i = 0; i < n_raw; i++) {
    const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
    const float intensity = ring[i] * z;             // :410 (uint16 -> int -> float multiply)
    const float range = sqrtf(x * x + y * y + z * z);  // utility.h:462-465
    if (range < P->lidar_min_range || range > P->lidar_max_range) continue;
    int rowIdn = ring[i];
    if (rowIdn < 0 || rowIdn >= NS) continue;
    if (rowIdn % P->downsample_rate != 0) continue;
    float horizonAngle = atan2f(x, y) * 180 / M_PI;  // :437 float atan2, float*int, then double divide, stored float
    int columnIdn = -round((horizonAngle - 90.0) / ang_res_x) + H / 2;  // :440
    if (columnIdn >= H) columnIdn -= H;
    if (columnIdn < 0 || columnIdn >= H) continue;
    if (range_mat[rowIdn * H + columnIdn] != FLT_MAX) continue;  // first point wins :451
    float sx = x, sy = y, sz = z;
    if (d && d->enabled && rel_time) deskew_point(d, rel_time[i], sx, sy, sz);  // :454 (the range and the pixel come from the raw point)
    range_mat[rowIdn * H + columnIdn] = range;
    float* fc = full_cloud + 4 * ((size_t)columnIdn + (size_t)rowIdn * H);
    fc[0] = sx; fc[1] = sy; fc[2] = sz; fc[3] = intensity;
  }
  int count = 0;
  for (int i = 0; i < NS; i++) {
    start_ring[i] = count - 1 + 5;
    for (int j = 0; j < H; j++) {
      if (range_mat[i * H + j] != FLT_MAX) {
        point_col_ind[count] = j;
        point_range[count] = range_mat[i * H + j];
        memcpy(extracted + 4 * (size_t)count, full_cloud + 4 * ((size_t)j + (size_t)i * H), 4 * sizeof(float));
        ++count;
      }
    }
    end_ring[i] = count - 1 - 5;
  }
  return count;
}

// pcl::VoxelGrid<PointXYZI>::applyFilter (downsample_all_data = true, no limits, min_points_per_voxel 0).
// The unstable std::sort on cell index is fixed to (cell, point index) order.
int orc_voxelgrid(const float* pts, int n, float leaf, float* out) {
  if (n <= 0) return 0;
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < n; i++) for (int d = 0; d < 3; d++) { float c = pts[4 * (size_t)i + d]; mn[d] = std::min(mn[d], c); mx[d] = std::max(mx[d], c); }
  int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1, dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) { memcpy(out, pts, sizeof(float) * 4 * (size_t)n); return n; }  // PCL: warn + copy through
  int min_b[3], max_b[3], div_b[3];
  for (int d = 0; d < 3; d++) { min_b[d] = (int)std::floor(mn[d] * inv); max_b[d] = (int)std::floor(mx[d] * inv); div_b[d] = max_b[d] - min_b[d] + 1; }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<std::pair<int, int>> iv(n);
  for (int i = 0; i < n; i++) {
    int ijk0 = (int)std::floor(pts[4 * (size_t)i] * inv) - min_b[0];
    int ijk1 = (int)std::floor(pts[4 * (size_t)i + 1] * inv) - min_b[1];
    int ijk2 = (int)std::floor(pts[4 * (size_t)i + 2] * inv) - min_b[2];
    iv[i] = {ijk0 * mul[0] + ijk1 * mul[1] + ijk2 * mul[2], i};
  }
  std::sort(iv.begin(), iv.end());
  int m = 0;
  for (int first = 0; first < n;) {
    int last = first + 1;
    while (last < n && iv[last].first == iv[first].first) last++;
    float sx = 0, sy = 0, sz = 0, si = 0;  // pcl::CentroidPoint: float accumulators (AccumulatorXYZ Vector3f, AccumulatorIntensity float)
    for (int li = first; li < last; li++) { const float* p = pts + 4 * (size_t)iv[li].second; sx += p[0]; sy += p[1]; sz += p[2]; si += p[3]; }
    float cnt = (float)(last - first);
    out[4 * (size_t)m] = sx / cnt; out[4 * (size_t)m + 1] = sy / cnt; out[4 * (size_t)m + 2] = sz / cnt; out[4 * (size_t)m + 3] = si / cnt;
    m++;
    first = last;
  }
  return m;
}

// featureExtraction.cpp:87-266
int orc_extract_features(const orc_front_params* P, const float* extracted, int n, const int32_t* col_in,
                         const float* range_in, const int32_t* start_ring, const int32_t* end_ring,
                         float* curvature_out, int32_t* picked_out, int32_t* label_out,
                         float* corner, int32_t* n_corner, float* surface, int32_t* n_surf) {
  const int G = 8;  // guard cells: the reference reads/writes up to index -6 / n+5 of its heap arrays (Q6); guards are 0 and discarded
  std::vector<float> curv_g(n + 2 * G, 0.0f), range_g(n + 2 * G, 0.0f);
  std::vector<int> picked_g(n + 2 * G, 0), label_g(n + 2 * G, 0), col_g(n + 2 * G, 0);
  float* curv = curv_g.data() + G; float* range = range_g.data() + G;
  int* picked = picked_g.data() + G; int* label = label_g.data() + G; int* col = col_g.data() + G;
  for (int i = 0; i < n; i++) { range[i] = range_in[i]; col[i] = col_in[i]; }
  struct Sm { float value; int ind; };
  std::vector<Sm> smooth_g(n + 2 * G, Sm{0.0f, 0});
  Sm* smooth = smooth_g.data() + G;

  // calculateSmoothness :87-110 — float sum in the written order
  for (int i = 5; i < n - 5; i++) {
    float diffRange = range[i - 5] + range[i - 4] + range[i - 3] + range[i - 2] + range[i - 1] - range[i] * 10
                    + range[i + 1] + range[i + 2] + range[i + 3] + range[i + 4] + range[i + 5];
    curv[i] = diffRange * diffRange;
    picked[i] = 0; label[i] = 0;
    smooth[i].value = curv[i]; smooth[i].ind = i;
  }
  // markOccludedPoints :112-150
  for (int i = 5; i < n - 6; ++i) {
    float depth1 = range[i], depth2 = range[i + 1];
    int columnDiff = std::abs(int(col[i + 1] - col[i]));
    if (columnDiff < 10) {
      if (depth1 - depth2 > 0.3) { for (int k = 0; k <= 5; k++) picked[i - k] = 1; }
      else if (depth2 - depth1 > 0.3) { for (int k = 1; k <= 6; k++) picked[i + k] = 1; }
    }
    float diff1 = std::abs(float(range[i - 1] - range[i]));
    float diff2 = std::abs(float(range[i + 1] - range[i]));
    if (diff1 > 0.02 * range[i] && diff2 > 0.02 * range[i]) picked[i] = 1;
  }
  // extractFeatures :153-266
  int nc = 0, ns = 0;
  std::vector<float> scan; scan.reserve(4 * (size_t)P->horizon_scan);
  std::vector<float> scan_ds(4 * (size_t)std::max(1, P->horizon_scan));
  for (int i = 0; i < P->n_scan; i++) {
    scan.clear();
    for (int j = 0; j < 6; j++) {
      int sp = (start_ring[i] * (6 - j) + end_ring[i] * j) / 6;
      int ep = (start_ring[i] * (5 - j) + end_ring[i] * (j + 1)) / 6 - 1;
      if (sp >= ep) continue;
      std::sort(smooth + sp, smooth + ep, [](const Sm& a, const Sm& b) { return a.value < b.value || (a.value == b.value && a.ind < b.ind); });
      int largestPickedNum = 0;
      for (int k = ep; k >= sp; k--) {
        int ind = smooth[k].ind;
        if (picked[ind] == 0 && curv[ind] > P->edge_threshold) {
          largestPickedNum++;
          if (largestPickedNum <= 20) { label[ind] = 1; memcpy(corner + 4 * (size_t)nc, extracted + 4 * (size_t)ind, 16); nc++; }
          else break;
          picked[ind] = 1;
          for (int l = 1; l <= 5; l++) { int cd = std::abs(int(col[ind + l] - col[ind + l - 1])); if (cd > 10) break; picked[ind + l] = 1; }
          for (int l = -1; l >= -5; l--) { int cd = std::abs(int(col[ind + l] - col[ind + l + 1])); if (cd > 10) break; picked[ind + l] = 1; }
        }
      }
      for (int k = sp; k <= ep; k++) {
        int ind = smooth[k].ind;
        if (picked[ind] == 0 && curv[ind] < P->surf_threshold) {
          label[ind] = -1; picked[ind] = 1;
          for (int l = 1; l <= 5; l++) { int cd = std::abs(int(col[ind + l] - col[ind + l - 1])); if (cd > 10) break; picked[ind + l] = 1; }
          for (int l = -1; l >= -5; l--) { int cd = std::abs(int(col[ind + l] - col[ind + l + 1])); if (cd > 10) break; picked[ind + l] = 1; }
        }
      }
      for (int k = sp; k <= ep; k++) if (label[k] <= 0) { const float* p = extracted + 4 * (size_t)k; scan.insert(scan.end(), p, p + 4); }
    }
    int m = (int)(scan.size() / 4);
    if ((size_t)m * 4 > scan_ds.size()) scan_ds.resize((size_t)m * 4);
    int md = orc_voxelgrid(scan.data(), m, P->odometry_surf_leaf_size, scan_ds.data());
    if (md > 0) { memcpy(surface + 4 * (size_t)ns, scan_ds.data(), sizeof(float) * 4 * (size_t)md); ns += md; }
  }
  for (int i = 0; i < n; i++) { if (curvature_out) curvature_out[i] = curv[i]; if (picked_out) picked_out[i] = picked[i]; if (label_out) label_out[i] = label[i]; }
  *n_corner = nc; *n_surf = ns;
  return 0;
}

// pcl::getTransformation (pcl/common/impl/eigen.hpp), float
void orc_get_transformation(float x, float y, float z, float roll, float pitch, float yaw, float* t) {
  float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll), F = std::sin(roll), DE = D * E, DF = D * F;
  t[0] = A * C; t[1] = A * DF - B * E; t[2] = B * F + A * DE; t[3] = x;
  t[4] = B * C; t[5] = A * E + B * DF; t[6] = B * DE - A * F; t[7] = y;
  t[8] = -D;    t[9] = C * F;          t[10] = C * E;         t[11] = z;
  t[12] = 0; t[13] = 0; t[14] = 0; t[15] = 1;
}
// Eigen::Affine3f::rotation() of the row-major 4x4 T (lidarOdometry.cpp:130, 474, 548)
void orc_affine3f_rotation(const float* T16, float* R9) {
  float L[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = T16[i * 4 + j];
  orc::rotation_of_affine3f(L, R9);
}
void orc_get_translation_and_euler(const float* t, float* o) {
  o[0] = t[3]; o[1] = t[7]; o[2] = t[11];
  o[3] = std::atan2(t[9], t[10]);
  o[4] = std::asin(-t[8]);
  o[5] = std::atan2(t[4], t[0]);
}

}  // extern "C"

namespace {
inline void mat4f_mul(const float* A, const float* B, float* C) {
  float r[16];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j]; r[i * 4 + j] = s; }
  memcpy(C, r, sizeof(r));
}
inline void affine_inverse_f(const float* T, float* out) {  // Eigen Transform<float,3,Affine>::inverse(): linear().inverse(), -inv*t
  float a[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = T[i * 4 + j];
  float c[3][3];
  c[0][0] = a[1][1] * a[2][2] - a[1][2] * a[2][1]; c[0][1] = a[0][2] * a[2][1] - a[0][1] * a[2][2]; c[0][2] = a[0][1] * a[1][2] - a[0][2] * a[1][1];
  c[1][0] = a[1][2] * a[2][0] - a[1][0] * a[2][2]; c[1][1] = a[0][0] * a[2][2] - a[0][2] * a[2][0]; c[1][2] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
  c[2][0] = a[1][0] * a[2][1] - a[1][1] * a[2][0]; c[2][1] = a[0][1] * a[2][0] - a[0][0] * a[2][1]; c[2][2] = a[0][0] * a[1][1] - a[0][1] * a[1][0];
  float det = a[0][0] * c[0][0] + a[0][1] * c[1][0] + a[0][2] * c[2][0];
  float inv = 1.0f / det;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) out[i * 4 + j] = c[i][j] * inv;
  }
  for (int i = 0; i < 3; i++) out[i * 4 + 3] = -(out[i * 4 + 0] * T[3] + out[i * 4 + 1] * T[7] + out[i * 4 + 2] * T[11]);
  out[12] = out[13] = out[14] = 0; out[15] = 1;
}
}  // namespace

extern "C" void orc_odom_increment(const float* front6, const float* back6, float* incre6) {  // imageProjection.cpp:293-299
  float F[16], B[16], Fi[16], I[16];
  orc_get_transformation(front6[0], front6[1], front6[2], front6[3], front6[4], front6[5], F);
  orc_get_transformation(back6[0], back6[1], back6[2], back6[3], back6[4], back6[5], B);
  affine_inverse_f(F, Fi);
  mat4f_mul(Fi, B, I);
  orc_get_translation_and_euler(I, incre6);
}

struct orc_odom {
  orc_params RP;
  float ct_lambda;
  bool first = true;
  double cloudTimeCur = 0, cloudTimeLast = 0;  // read before assignment in the reference; restated as 0
  double lastOdomTime = -1;
  double lastMappingInterval = 9999.0;  // lidarOdometry.cpp:419
  float lidarMappingAffine[16];
  float transformation_interpolated[16];
  double Rotation[9], Translation[3], TranslationOld[3];
  float LaserOdomPose[6] = {0, 0, 0, 0, 0, 0};
  std::vector<float> featureOld;  // n*4
};

extern "C" {

orc_odom* orc_odom_create(const orc_params* rp, float ct_lambda) {
  orc_odom* o = new orc_odom();
  o->RP = *rp; o->ct_lambda = ct_lambda;
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  memcpy(o->lidarMappingAffine, I, sizeof(I)); memcpy(o->transformation_interpolated, I, sizeof(I));
  for (int i = 0; i < 9; i++) o->Rotation[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; i++) o->Translation[i] = o->TranslationOld[i] = 0;
  return o;
}
void orc_odom_destroy(orc_odom* o) { delete o; }
void orc_odom_backend_odometry(orc_odom* o, double stamp) { o->lastOdomTime = stamp; }

static void update_transform(orc_odom* o) {  // lidarOdometry.cpp:572-626 (pose part)
  float step[16];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) step[i * 4 + j] = (float)o->Rotation[i * 3 + j]; step[i * 4 + 3] = (float)o->Translation[i]; }
  step[12] = step[13] = step[14] = 0; step[15] = 1;
  float pose[16], inv[16], tp[16];
  orc_get_transformation(o->LaserOdomPose[0], o->LaserOdomPose[1], o->LaserOdomPose[2], o->LaserOdomPose[3], o->LaserOdomPose[4], o->LaserOdomPose[5], pose);
  affine_inverse_f(step, inv);
  mat4f_mul(pose, inv, tp);
  memcpy(o->lidarMappingAffine, step, sizeof(step));
  orc_get_translation_and_euler(tp, o->LaserOdomPose);
  for (int i = 0; i < 3; i++) o->TranslationOld[i] = o->Translation[i];
}

int orc_odom_cloud(orc_odom* o, double stamp, const float* corner, int n_corner, const float* surface, int n_surf,
                   float* pose6, double* rot9, double* trans3) {
  o->cloudTimeCur = stamp;
  std::vector<float> featureLast((size_t)(n_corner + n_surf) * 4);
  if (n_corner) memcpy(featureLast.data(), corner, sizeof(float) * 4 * (size_t)n_corner);
  if (n_surf) memcpy(featureLast.data() + 4 * (size_t)n_corner, surface, sizeof(float) * 4 * (size_t)n_surf);
  int ret;
  if (o->first) {
    o->first = false;
    o->featureOld = featureLast;
    ret = 0;
  } else if (o->lastOdomTime == -1.0) {  // Q4: gated until the back end has published once
    update_transform(o);
    o->featureOld = featureLast;
    ret = 1;
  } else {
    double latestInterval = o->cloudTimeCur - o->cloudTimeLast;
    // stateLinearPropagation :700-712 — translation of the last step scaled by the interval ratio, rotation zeroed
    double ratio = latestInterval / o->lastMappingInterval;
    float v[6]; orc_get_translation_and_euler(o->lidarMappingAffine, v);
    v[3] = v[4] = v[5] = 0;
    for (int i = 0; i < 6; i++) v[i] *= (float)ratio;
    orc_get_transformation(v[0], v[1], v[2], v[3], v[4], v[5], o->transformation_interpolated);
    o->cloudTimeLast = o->cloudTimeCur;
    o->lastMappingInterval = latestInterval;
    // scanRegeistration :448-501
    const int nOld = (int)(o->featureOld.size() / 4);
    std::vector<float> propagated(o->featureOld.size());
    orc_transform_cloud_f(o->featureOld.data(), propagated.data(), nOld, 4, o->transformation_interpolated);
    orc_reg* reg = orc_reg_create(&o->RP);
    int rc = orc_reg_set_target(reg, featureLast.data(), n_corner + n_surf, 4);
    if (!rc) rc = orc_reg_set_source(reg, propagated.data(), nOld, 4);
    float Tf[16];
    if (!rc) rc = orc_reg_align(reg, nullptr, Tf, nullptr, nullptr, nullptr);
    if (rc < 0) { orc_reg_destroy(reg); return rc; }
    mat4f_mul(o->transformation_interpolated, Tf, o->transformation_interpolated);
    {   // :474-475: Rotation = transformation_interpolated.rotation().cast<double>() — Eigen's float polar factor (orc_linalg.hpp)
      float L[9], Rf[9];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = o->transformation_interpolated[i * 4 + j];
      orc::rotation_of_affine3f(L, Rf);
      for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o->Rotation[i * 3 + j] = (double)Rf[i * 3 + j]; o->Translation[i] = (double)o->transformation_interpolated[i * 4 + 3]; }
    }
    double reg_t[3] = {0, 0, 0};
    rc = orc_reg_compute_translation(reg, reg_t, o->Translation, o->TranslationOld, 0.1, 0.1, o->ct_lambda, nullptr);
    orc_reg_destroy(reg);
    if (rc < 0) return rc;
    for (int i = 0; i < 3; i++) o->Translation[i] += reg_t[i];
    update_transform(o);
    o->featureOld = featureLast;
    ret = 2;
  }
  if (pose6) memcpy(pose6, o->LaserOdomPose, sizeof(float) * 6);
  if (rot9) memcpy(rot9, o->Rotation, sizeof(double) * 9);
  if (trans3) memcpy(trans3, o->Translation, sizeof(double) * 3);
  return ret;
}

}  // extern "C"
