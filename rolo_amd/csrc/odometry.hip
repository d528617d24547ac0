// K13 — the per-frame odometry driver (host side).
// Replaces LidarOdometry::cloudHandler / stateLinearPropagation / scanRegeistration / updateTransform
// (reference src/lidarOdometry.cpp:503-570, 700-712, 448-501, 572-626) on feature clouds, i.e. everything of the
// rolo_lidarOdometry node between "fromROSMsg" and "publish". The registration itself is the HIP path
// (rolo_register_async / rolo_register_wait: both LM stages enqueued back to back, one host wait per frame).
// The fp32 pose algebra restates pcl::getTransformation / getTranslationAndEulerAngles / Eigen::Affine3f products
// (PCL, Eigen: not vendored by the reference).
#include "rolo_internal.hpp"
#include "polar_f32.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace rolo {
void ctx_set_error(const char* msg);
int ctx_device(rolo_ctx* c);
void ctx_set_fused_lm(rolo_ctx* c, int on);
int ctx_create_high_priority(int device, rolo_ctx** out);
int ctx_set_pair_device(rolo_ctx* c, const float* d_src, int n_src, int stride_src, const float* T16_host_or_null, const float* d_tgt, int n_tgt, int stride_tgt);
}

namespace {

struct Aff { float m[16]; };  // row-major 4x4

Aff aff_identity() { Aff a; for (int i = 0; i < 16; i++) a.m[i] = (i % 5 == 0) ? 1.f : 0.f; return a; }

// pcl::getTransformation(x, y, z, roll, pitch, yaw)
Aff get_transformation(float x, float y, float z, float roll, float pitch, float yaw) {
  const float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll), F = std::sin(roll);
  const float DE = D * E, DF = D * F;
  Aff t;
  t.m[0] = A * C; t.m[1] = A * DF - B * E; t.m[2] = B * F + A * DE; t.m[3] = x;
  t.m[4] = B * C; t.m[5] = A * E + B * DF; t.m[6] = B * DE - A * F; t.m[7] = y;
  t.m[8] = -D;    t.m[9] = C * F;          t.m[10] = C * E;         t.m[11] = z;
  t.m[12] = 0; t.m[13] = 0; t.m[14] = 0; t.m[15] = 1;
  return t;
}
// pcl::getTranslationAndEulerAngles
void get_translation_and_euler(const Aff& t, float* o) {
  o[0] = t.m[3]; o[1] = t.m[7]; o[2] = t.m[11];
  o[3] = std::atan2(t.m[9], t.m[10]);
  o[4] = std::asin(-t.m[8]);
  o[5] = std::atan2(t.m[4], t.m[0]);
}
Aff aff_mul(const Aff& a, const Aff& b) {
  Aff c;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0; for (int k = 0; k < 4; k++) s += a.m[i * 4 + k] * b.m[k * 4 + j]; c.m[i * 4 + j] = s; }
  return c;
}
// Eigen::Transform<float,3,Affine>::inverse(): cofactor inverse of the linear part, translation -inv * t
Aff aff_inverse(const Aff& T) {
  const float* a = T.m;
  float c[9];
  c[0] = a[5] * a[10] - a[6] * a[9]; c[1] = a[2] * a[9] - a[1] * a[10]; c[2] = a[1] * a[6] - a[2] * a[5];
  c[3] = a[6] * a[8] - a[4] * a[10]; c[4] = a[0] * a[10] - a[2] * a[8]; c[5] = a[2] * a[4] - a[0] * a[6];
  c[6] = a[4] * a[9] - a[5] * a[8];  c[7] = a[1] * a[8] - a[0] * a[9];  c[8] = a[0] * a[5] - a[1] * a[4];
  const float det = a[0] * c[0] + a[1] * c[3] + a[2] * c[6];
  const float inv = 1.0f / det;
  Aff o;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.m[i * 4 + j] = c[i * 3 + j] * inv;
  for (int i = 0; i < 3; i++) o.m[i * 4 + 3] = -(o.m[i * 4] * a[3] + o.m[i * 4 + 1] * a[7] + o.m[i * 4 + 2] * a[11]);
  o.m[12] = o.m[13] = o.m[14] = 0; o.m[15] = 1;
  return o;
}

}  // namespace

struct rolo_odom {
  rolo_ctx* ctx;
  float ct_lambda;
  bool first = true;                       // isFirstFrame
  double cloudTimeCur = 0, cloudTimeLast = 0;  // cloudTimeLast is read before its first assignment in the reference: 0 here
  double lastOdomTime = -1;                // lidarOdometry.cpp:417
  double lastMappingInterval = 9999.0;     // :419
  Aff lidarMappingAffine = aff_identity();
  Aff transformation_interpolated = aff_identity();
  double Rotation[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Translation[3] = {0, 0, 0}, TranslationOld[3] = {0, 0, 0};
  float LaserOdomPose[6] = {0, 0, 0, 0, 0, 0};
  std::vector<float> featureOld;           // n x 4 (rolo_odom_cloud: host-side hand-over)
  // rolo_odom_submit / _collect / _frame: device-resident hand-over. K1-K4 run on their own context (stream + buffers),
  // so the features of frame k+1 are extracted while frame k registers on `ctx`.
  rolo_ctx* fctx = nullptr;
  float4* d_feat[3] = {nullptr, nullptr, nullptr};  // ring: [old] = previous features, [old+1], [old+2] = submitted frames
  float4* d_prop = nullptr;
  size_t d_cap = 0;
  int old_buf = 0, nOld = 0;
  int nCornerOld = 0;   // corners lead the feature cloud of the last collected frame
  struct Slot { double stamp = 0; int* h_counts = nullptr; hipEvent_t done = nullptr; } q[2];
  int q_head = 0, q_len = 0;
  bool reuse_cov = false, cov_chain = false;  // cov_chain: the context's target covariances belong to d_feat[old_buf]
  // Early source (frame-at-a-time use of submit / collect / frame): the source of a registration is the PREVIOUS frame's features moved by the
  // forward prediction — known when the new frame is submitted, long before its features exist. Its neighbour search then runs on the
  // registration stream while K1-K4 of the new frame run on the front-end stream; collect only has the target left to search.
  bool early_src = false; double early_stamp = 0; Aff early_T = aff_identity();
  bool early_source_enabled = false;  // ROLO_ODOM_EARLY_SOURCE (off: measured slower, see rolo_hip.h)
  int fused_lm = 2;       // ROLO_ODOM_FUSED_LM: rolo_params.fused_lm of this driver's registrations (one frame at a time: the shortest chain — 2, one launch per frame, since round 6; 1, one launch per trial, before)
  rolo_stats last_rot{}, last_trans{};
};

namespace {
// Rotation = transformation_interpolated.rotation().cast<double>(); Translation = ....translation().cast<double>() (lidarOdometry.cpp:474-475).
// Affine3f::rotation() is Eigen's float polar factor of the linear part (polar_f32.hpp), not the linear part itself.
void set_rotation_translation(rolo_odom* o) {
  float L[9], R[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = o->transformation_interpolated.m[i * 4 + j];
  rolo::polar::rotation_f32(L, R);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) o->Rotation[i * 3 + j] = (double)R[i * 3 + j]; o->Translation[i] = (double)o->transformation_interpolated.m[i * 4 + 3]; }
}
void update_transform(rolo_odom* o) {  // lidarOdometry.cpp:572-626, pose part
  Aff step;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) step.m[i * 4 + j] = (float)o->Rotation[i * 3 + j]; step.m[i * 4 + 3] = (float)o->Translation[i]; }
  step.m[12] = step.m[13] = step.m[14] = 0; step.m[15] = 1;
  const Aff pose = get_transformation(o->LaserOdomPose[0], o->LaserOdomPose[1], o->LaserOdomPose[2], o->LaserOdomPose[3], o->LaserOdomPose[4], o->LaserOdomPose[5]);
  const Aff moved = aff_mul(pose, aff_inverse(step));
  o->lidarMappingAffine = step;
  get_translation_and_euler(moved, o->LaserOdomPose);
  for (int i = 0; i < 3; i++) o->TranslationOld[i] = o->Translation[i];
}
}  // namespace

extern "C" {

int rolo_odom_create(rolo_ctx* ctx, float ct_lambda, rolo_odom** out) {
  if (!ctx || !out) return ROLO_EINVAL;
  rolo_odom* o = new rolo_odom();
  o->ctx = ctx; o->ct_lambda = ct_lambda;
  if (const char* e = getenv("ROLO_ODOM_EARLY_SOURCE")) o->early_source_enabled = atoi(e) != 0;   // A/B runs
  // (rolo_params.fused_lm is NOT switched behind the caller's back here: the driver asserts its own option right before every registration it
  // enqueues — a later rolo_set_params with the caller's own parameter block cannot silently revert it, nor does creating a driver change
  // what a plain rolo_register_async on the same context does afterwards beyond the frames the driver itself runs)
  *out = o;
  return ROLO_OK;
}
void rolo_odom_destroy(rolo_odom* o) {
  if (!o) return;
  if (o->fctx) rolo_ctx_destroy(o->fctx);  // synchronises its stream
  for (float4* b : o->d_feat) if (b) (void)hipFree(b);
  if (o->d_prop) (void)hipFree(o->d_prop);
  for (auto& sl : o->q) { if (sl.h_counts) (void)hipHostFree(sl.h_counts); if (sl.done) (void)hipEventDestroy(sl.done); }
  delete o;
}

void rolo_affine3f_rotation(const float* T16, float* R9) {   // lidarOdometry.cpp:474
  if (!T16 || !R9) return;
  float L[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = T16[i * 4 + j];
  rolo::polar::rotation_f32(L, R9);
}
void rolo_odom_increment(const float* front6, const float* back6, float* incre6) {  // imageProjection.cpp:345-351
  const Aff F = get_transformation(front6[0], front6[1], front6[2], front6[3], front6[4], front6[5]);
  const Aff B = get_transformation(back6[0], back6[1], back6[2], back6[3], back6[4], back6[5]);
  get_translation_and_euler(aff_mul(aff_inverse(F), B), incre6);
}

static int ensure_front_ctx(rolo_odom* o) {
  if (o->fctx) return ROLO_OK;
  static const bool plain = [] { const char* e = getenv("ROLO_ODOM_FRONT_PRIORITY"); return e && atoi(e) == 0; }();   // A/B: 0 = a normal-priority front-end stream (round 2)
  int rc = plain ? rolo_ctx_create(rolo::ctx_device(o->ctx), &o->fctx) : rolo::ctx_create_high_priority(rolo::ctx_device(o->ctx), &o->fctx);
  if (rc) return rc;
  for (auto& sl : o->q) {
    if (hipHostMalloc((void**)&sl.h_counts, 4 * sizeof(int)) != hipSuccess || hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) {
      rolo::ctx_set_error("pinned buffer / event creation failed"); return ROLO_EHIP;
    }
  }
  return ROLO_OK;
}

int rolo_odom_set_deskew(rolo_odom* o, const rolo_deskew* d, const float* rel_time, int n_raw, int on_device) {
  if (!o) return ROLO_EINVAL;
  if (hipSetDevice(rolo::ctx_device(o->ctx)) != hipSuccess) { rolo::ctx_set_error("hipSetDevice failed"); return ROLO_EHIP; }
  const int rc = ensure_front_ctx(o);
  if (rc) return rc;
  return rolo_front_set_deskew(o->fctx, d, rel_time, n_raw, on_device);
}

int rolo_odom_set_option(rolo_odom* o, int option, int value) {
  if (!o) return ROLO_EINVAL;
  if (option == ROLO_ODOM_REUSE_COVARIANCES) { o->reuse_cov = value != 0; o->cov_chain = false; return ROLO_OK; }
  if (option == ROLO_ODOM_FUSED_LM) { if (value < 0 || value > 2) return ROLO_EINVAL; o->fused_lm = value; return ROLO_OK; }
  if (option == ROLO_ODOM_EARLY_SOURCE) { o->early_source_enabled = value != 0; return ROLO_OK; }
  return ROLO_EINVAL;
}

int rolo_odom_backend_odometry(rolo_odom* o, double stamp) {  // odometryHandler :440-446
  if (!o) return ROLO_EINVAL;
  o->lastOdomTime = stamp;
  return ROLO_OK;
}

int rolo_odom_cloud(rolo_odom* o, double stamp, const float* corner, int n_corner, const float* surface, int n_surf,
                    float* pose6, double* rot9, double* trans3) {
  if (!o || n_corner < 0 || n_surf < 0 || (n_corner && !corner) || (n_surf && !surface)) return ROLO_EINVAL;
  o->cloudTimeCur = stamp;
  std::vector<float> featureLast((size_t)(n_corner + n_surf) * 4);  // *featureLast = *CloudCornerLast + *CloudSurfLast
  if (n_corner) memcpy(featureLast.data(), corner, sizeof(float) * 4 * (size_t)n_corner);
  if (n_surf) memcpy(featureLast.data() + 4 * (size_t)n_corner, surface, sizeof(float) * 4 * (size_t)n_surf);
  int ret;
  if (o->first) {
    o->first = false;
    o->featureOld.swap(featureLast);
    ret = 0;
  } else if (o->lastOdomTime == -1.0) {  // no scan matching until the back end has published once
    update_transform(o);
    o->featureOld.swap(featureLast);
    ret = 1;
  } else {
    const double latestInterval = o->cloudTimeCur - o->cloudTimeLast;
    // stateLinearPropagation :700-712
    const double ratio = latestInterval / o->lastMappingInterval;
    float v[6];
    get_translation_and_euler(o->lidarMappingAffine, v);
    v[3] = v[4] = v[5] = 0;
    for (int i = 0; i < 6; i++) v[i] *= (float)ratio;
    o->transformation_interpolated = get_transformation(v[0], v[1], v[2], v[3], v[4], v[5]);
    o->cloudTimeLast = o->cloudTimeCur;
    o->lastMappingInterval = latestInterval;
    // scanRegeistration :448-501
    const int nOld = (int)(o->featureOld.size() / 4);
    std::vector<float> propagated(o->featureOld.size());
    int rc = rolo_transform_cloud(o->ctx, o->featureOld.data(), propagated.data(), nOld, 4, o->transformation_interpolated.m);
    if (rc) return rc;
    if ((rc = rolo_set_target(o->ctx, featureLast.data(), n_corner + n_surf, 4))) return rc;
    if ((rc = rolo_set_source(o->ctx, propagated.data(), nOld, 4))) return rc;
    double guess_t[3];  // Translation after the rotation stage = translation of T_interp * T_rot = that of T_interp
    for (int i = 0; i < 3; i++) guess_t[i] = (double)o->transformation_interpolated.m[i * 4 + 3];
    const double zero3[3] = {0, 0, 0};
    rolo::ctx_set_fused_lm(o->ctx, o->fused_lm);
    if ((rc = rolo_register_async(o->ctx, nullptr, zero3, guess_t, o->TranslationOld, 0.1, 0.1, o->ct_lambda))) return rc;
    float Tf[16]; double reg_t[3];
    if ((rc = rolo_register_wait(o->ctx, Tf, nullptr, reg_t, &o->last_rot, &o->last_trans))) return rc;
    Aff step; memcpy(step.m, Tf, sizeof(Tf));
    o->transformation_interpolated = aff_mul(o->transformation_interpolated, step);  // :472
    set_rotation_translation(o);   // :474-475
    for (int i = 0; i < 3; i++) o->Translation[i] += reg_t[i];  // :500
    update_transform(o);
    o->featureOld.swap(featureLast);
    ret = 2;
  }
  if (pose6) memcpy(pose6, o->LaserOdomPose, sizeof(float) * 6);
  if (rot9) memcpy(rot9, o->Rotation, sizeof(double) * 9);
  if (trans3) memcpy(trans3, o->Translation, sizeof(double) * 3);
  return ret;
}

// stateLinearPropagation :700-712 for a frame stamped `stamp`, from the driver's current state (does not change it)
static Aff predicted_transform(const rolo_odom* o, double stamp) {
  const double latestInterval = stamp - o->cloudTimeLast;
  const double ratio = latestInterval / o->lastMappingInterval;
  float v[6];
  get_translation_and_euler(o->lidarMappingAffine, v);
  v[3] = v[4] = v[5] = 0;
  for (int i = 0; i < 6; i++) v[i] *= (float)ratio;
  return get_transformation(v[0], v[1], v[2], v[3], v[4], v[5]);
}

static int submit_common(rolo_odom* o, const rolo_front_params* P, double stamp, const void* pts, int stride, const uint16_t* ring,
                         const rolo_cloud_layout* layout, int n_raw, int on_device) {
  if (o->q_len == 2) { rolo::ctx_set_error("two frames are already in flight: collect one first"); return ROLO_ESTATE; }
  if (hipSetDevice(rolo::ctx_device(o->ctx)) != hipSuccess) { rolo::ctx_set_error("hipSetDevice failed"); return ROLO_EHIP; }
  int rc;
  if ((rc = ensure_front_ctx(o))) return rc;
  const size_t cap = rolo::front_feature_capacity(P);
  if (cap > o->d_cap) {
    if (o->nOld > 0 || o->q_len > 0) { rolo::ctx_set_error("front parameters grew between frames"); return ROLO_ESTATE; }
    float4** bufs[4] = {&o->d_feat[0], &o->d_feat[1], &o->d_feat[2], &o->d_prop};
    for (float4** b : bufs) {
      if (*b) { (void)hipFree(*b); *b = nullptr; }
      if (hipMalloc((void**)b, sizeof(float4) * cap) != hipSuccess) { rolo::ctx_set_error("hipMalloc failed (odometry feature buffers)"); return ROLO_EHIP; }
    }
    o->d_cap = cap;
  }
  const int buf = (o->old_buf + 1 + o->q_len) % 3;
  rolo_odom::Slot& sl = o->q[(o->q_head + o->q_len) % 2];
  if (layout) rc = rolo::front_frame_features_from_msg(o->fctx, P, static_cast<const unsigned char*>(pts), layout, n_raw, on_device != 0, o->d_feat[buf], sl.h_counts, sl.done);
  else rc = rolo::front_frame_features_enqueue(o->fctx, P, static_cast<const float*>(pts), stride, ring, n_raw, on_device != 0, o->d_feat[buf], sl.h_counts, sl.done);
  if (rc) return rc;
  sl.stamp = stamp;
  // nothing else in flight and the next collect will register: start on the source now (see early_src)
  o->early_src = false;
  if (o->early_source_enabled && o->q_len == 0 && !o->first && o->lastOdomTime != -1.0 && o->nOld > 0 && !(o->reuse_cov && o->cov_chain)) {
    const Aff T = predicted_transform(o, stamp);
    hipStream_t s = (hipStream_t)rolo_ctx_stream(o->ctx);
    float4* d_featOld = o->d_feat[o->old_buf];
    if (rolo::launch_transform_cloud(reinterpret_cast<const float*>(d_featOld), reinterpret_cast<float*>(o->d_prop), o->nOld, 4, nullptr, T.m, s) == hipSuccess &&
        rolo_set_source_device(o->ctx, reinterpret_cast<const float*>(o->d_prop), o->nOld, 4) == ROLO_OK &&
        rolo_compute_covariances(o->ctx) == ROLO_OK) {   // the context's target (the previous frame's) keeps its covariances: only the source is searched, asynchronously
      o->early_src = true; o->early_stamp = stamp; o->early_T = T;
    }
  }
  o->q_len++;
  return ROLO_OK;
}

int rolo_odom_submit(rolo_odom* o, const rolo_front_params* P, double stamp, const float* pts, int stride, const uint16_t* ring, int n_raw,
                     int pts_on_device) {
  if (!o || !P || !pts || !ring || stride < 3 || n_raw < 0) return ROLO_EINVAL;
  return submit_common(o, P, stamp, pts, stride, ring, nullptr, n_raw, pts_on_device);
}

int rolo_odom_submit_msg(rolo_odom* o, const rolo_front_params* P, double stamp, const uint8_t* data, const rolo_cloud_layout* layout, int n_points,
                         int data_on_device) {
  if (!o || !P || !data || !layout || n_points < 0) return ROLO_EINVAL;
  return submit_common(o, P, stamp, data, 0, nullptr, layout, n_points, data_on_device);
}

int rolo_odom_collect(rolo_odom* o, float* pose6, double* rot9, double* trans3, int* counts3) {
  if (!o) return ROLO_EINVAL;
  if (o->q_len == 0) { rolo::ctx_set_error("no submitted frame to collect"); return ROLO_ESTATE; }
  if (hipSetDevice(rolo::ctx_device(o->ctx)) != hipSuccess) { rolo::ctx_set_error("hipSetDevice failed"); return ROLO_EHIP; }
  rolo_odom::Slot& sl = o->q[o->q_head];
  o->q_head = (o->q_head + 1) % 2; o->q_len--;
  if (hipEventSynchronize(sl.done) != hipSuccess) { rolo::ctx_set_error("front-end stream failed"); return ROLO_EHIP; }
  int counts[3] = {sl.h_counts[0], sl.h_counts[1], sl.h_counts[2]};
  if (counts3) memcpy(counts3, counts, sizeof(counts));
  const int buf = (o->old_buf + 1) % 3;
  float4* d_featNew = o->d_feat[buf];
  float4* d_featOld = o->d_feat[o->old_buf];
  const int nNew = counts[1] + counts[2];
  o->cloudTimeCur = sl.stamp;
  int ret, rc;
  if (o->first) {
    o->first = false;
    ret = 0;
  } else if (o->lastOdomTime == -1.0) {
    update_transform(o);
    ret = 1;
  } else {
    const double latestInterval = o->cloudTimeCur - o->cloudTimeLast;
    o->transformation_interpolated = predicted_transform(o, o->cloudTimeCur);  // stateLinearPropagation :700-712
    o->cloudTimeLast = o->cloudTimeCur;
    o->lastMappingInterval = latestInterval;
    // scanRegeistration :448-501 on device-resident clouds
    hipStream_t s = (hipStream_t)rolo_ctx_stream(o->ctx);
    const bool early = o->early_src && o->early_stamp == o->cloudTimeCur && memcmp(o->early_T.m, o->transformation_interpolated.m, sizeof(o->early_T.m)) == 0;
    o->early_src = false;
    const bool pair_pack = !early && !(o->reuse_cov && o->cov_chain) && o->nOld > 0 && nNew > 0;
    if (pair_pack) {   // *Propagated_cloud = T * featureOld (:459) and both setInput* as ONE launch: the transform rides on the pack
      if ((rc = rolo::ctx_set_pair_device(o->ctx, reinterpret_cast<const float*>(d_featOld), o->nOld, 4, o->transformation_interpolated.m,
                                          reinterpret_cast<const float*>(d_featNew), nNew, 4))) return rc;
    } else if (!early) {   // (early: the propagated source is in d_prop already and its covariances are on their way)
      if (o->nOld > 0 && rolo::launch_transform_cloud(reinterpret_cast<const float*>(d_featOld), reinterpret_cast<float*>(o->d_prop), o->nOld, 4, nullptr,
                                                      o->transformation_interpolated.m, s) != hipSuccess) {
        rolo::ctx_set_error("transform kernel launch failed"); return ROLO_EHIP;
      }
      if ((rc = rolo_set_source_device(o->ctx, reinterpret_cast<const float*>(o->d_prop), o->nOld, 4))) return rc;
      if (o->reuse_cov && o->cov_chain) { if ((rc = rolo_adopt_target_covariances(o->ctx))) return rc; }
    }
    if (!pair_pack && (rc = rolo_set_target_device(o->ctx, reinterpret_cast<const float*>(d_featNew), nNew, 4))) return rc;
    double guess_t[3];
    for (int i = 0; i < 3; i++) guess_t[i] = (double)o->transformation_interpolated.m[i * 4 + 3];
    const double zero3[3] = {0, 0, 0};
    o->cov_chain = false;
    rolo::ctx_set_fused_lm(o->ctx, o->fused_lm);
    if ((rc = rolo_register_async(o->ctx, nullptr, zero3, guess_t, o->TranslationOld, 0.1, 0.1, o->ct_lambda))) return rc;
    float Tf[16]; double reg_t[3];
    if ((rc = rolo_register_wait(o->ctx, Tf, nullptr, reg_t, &o->last_rot, &o->last_trans))) return rc;
    o->cov_chain = true;  // the context now holds the covariances of d_featNew as its target's
    Aff step; memcpy(step.m, Tf, sizeof(Tf));
    o->transformation_interpolated = aff_mul(o->transformation_interpolated, step);  // :472
    set_rotation_translation(o);   // :474-475
    for (int i = 0; i < 3; i++) o->Translation[i] += reg_t[i];  // :500
    update_transform(o);
    ret = 2;
  }
  o->old_buf = buf;
  o->nOld = nNew;
  o->nCornerOld = counts[1];
  if (pose6) memcpy(pose6, o->LaserOdomPose, sizeof(float) * 6);
  if (rot9) memcpy(rot9, o->Rotation, sizeof(double) * 9);
  if (trans3) memcpy(trans3, o->Translation, sizeof(double) * 3);
  return ret;
}

int rolo_odom_get_features(rolo_odom* o, float* features, int cap_points, int* n_corner, int* n_surface) {
  if (!o || !n_corner || !n_surface) return ROLO_EINVAL;
  *n_corner = o->nCornerOld; *n_surface = o->nOld - o->nCornerOld;
  if (!features || o->nOld == 0) return ROLO_OK;
  if (cap_points < o->nOld) { rolo::ctx_set_error("feature buffer too small"); return ROLO_EINVAL; }
  if (hipSetDevice(rolo::ctx_device(o->ctx)) != hipSuccess) { rolo::ctx_set_error("hipSetDevice failed"); return ROLO_EHIP; }
  // the last collected frame's features sit in d_feat[old_buf]; the front-end stream finished writing them before collect returned
  if (hipMemcpy(features, o->d_feat[o->old_buf], sizeof(float4) * (size_t)o->nOld, hipMemcpyDeviceToHost) != hipSuccess) { rolo::ctx_set_error("feature read-back failed"); return ROLO_EHIP; }
  return ROLO_OK;
}

int rolo_odom_frame(rolo_odom* o, const rolo_front_params* P, double stamp, const float* pts, int stride, const uint16_t* ring, int n_raw,
                    int pts_on_device, float* pose6, double* rot9, double* trans3, int* counts3) {
  if (!o) return ROLO_EINVAL;
  if (o->q_len != 0) { rolo::ctx_set_error("rolo_odom_frame with submitted frames pending: collect them first"); return ROLO_ESTATE; }
  const int rc = rolo_odom_submit(o, P, stamp, pts, stride, ring, n_raw, pts_on_device);
  if (rc) return rc;
  return rolo_odom_collect(o, pose6, rot9, trans3, counts3);
}

}  // extern "C"
