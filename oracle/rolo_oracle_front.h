/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header). PARITY UNPINNED by the reference itself.
 *
 * CPU restatement of the front half of the hot path: range-image projection (src/imageProjection.cpp),
 * curvature feature extraction (src/featureExtraction.cpp) and the per-frame odometry driver
 * (src/lidarOdometry.cpp:325-713). Citations are relative to /root/reference.
 */
#ifndef ROLO_ORACLE_FRONT_H
#define ROLO_ORACLE_FRONT_H
#include <stdint.h>
#include "rolo_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_front_params {
  int n_scan;              /* utility.h:310 N_SCAN */
  int horizon_scan;        /* :311 Horizon_SCAN */
  int downsample_rate;     /* :312 */
  float lidar_min_range;   /* :313 */
  float lidar_max_range;   /* :314 */
  float edge_threshold;    /* :318 */
  float surf_threshold;    /* :319 */
  float odometry_surf_leaf_size; /* :323 */
} orc_front_params;
void orc_front_default_params(orc_front_params* p); /* config/params.yaml values */

/* K1 + K2: imageProjection.cpp:399-505 (deskew off — every shipped config).
 * in: n_raw points, xyz at float offsets 0..2 of `stride`-float records, ring[n_raw] (uint16).
 * out: range_mat[n_scan*H] (FLT_MAX = empty), full_cloud[n_scan*H*4] (x,y,z,intensity; 0 where empty),
 *      extracted[N*4], point_col_ind[N], point_range[N], start_ring[n_scan], end_ring[n_scan].
 * Arrays sized for N = n_scan*H. Returns N (valid points) or <0. */
int orc_project(const orc_front_params* P, const float* pts, int stride, const uint16_t* ring, int n_raw,
                float* range_mat, float* full_cloud, float* extracted, int32_t* point_col_ind,
                float* point_range, int32_t* start_ring, int32_t* end_ring);

/* The same with ImageProjection::deskewPoint (imageProjection.cpp:368-396) for clouds that carry a per-point time
 * (timeFlag == 1, deskewCloudInfo :330-361): rel_time[i] = fabs(point.time) is what :358-359 stores in
 * deskewCloud->points[i].intensity; odom_incre_rpy / odom_time_diff are odomIncreRoll/Pitch/Yaw and odomTimeDiff of
 * :349-351. d == NULL or !d->enabled: no de-skew (deskewPoint returns the point, :371-372). */
typedef struct orc_deskew { int enabled; float odom_incre_rpy[3]; float scan_period; double odom_time_diff; } orc_deskew;
int orc_project_deskew(const orc_front_params* P, const float* pts, int stride, const uint16_t* ring, int n_raw,
                       const float* rel_time, const orc_deskew* d,
                       float* range_mat, float* full_cloud, float* extracted, int32_t* point_col_ind,
                       float* point_range, int32_t* start_ring, int32_t* end_ring);
/* deskewCloudInfo for clouds WITHOUT a time field (timeFlag == -1, imageProjection.cpp:270-327): the per-point time
 * interpolated from the azimuth, rel_time[i] = scanPeriod * relTime — what that branch stores in deskewCloud->points[i].intensity. */
void orc_azimuth_times(const float* pts, int stride, int n, float scan_period, float* rel_time);
/* lidarOdomAffineFront.inverse() * lidarOdomAffineBack -> getTranslationAndEulerAngles (:293-299, 345-351); poses as
 * x, y, z, roll, pitch, yaw (odom2affine :514-522 goes through tf's quaternion -> RPY, done by the caller) */
void orc_odom_increment(const float* front6, const float* back6, float* incre6);

/* K3 + K4: featureExtraction.cpp:87-266. Per-frame arrays are zero-initialised and the unstable
 * std::sort tie order is fixed to (curvature, index) (Q7).
 * out: curvature[N], neighbor_picked[N], label[N] (after extraction), corner[*n_corner*4], surface[*n_surf*4]
 * (surface = per-ring pcl::VoxelGrid output appended in ring order). corner/surface sized for N points. */
int orc_extract_features(const orc_front_params* P, const float* extracted, int n, const int32_t* point_col_ind,
                         const float* point_range, const int32_t* start_ring, const int32_t* end_ring,
                         float* curvature, int32_t* neighbor_picked, int32_t* label,
                         float* corner, int32_t* n_corner, float* surface, int32_t* n_surf);

/* pcl::VoxelGrid<PointXYZI>::filter (featureExtraction.cpp:58,257-258); pts n*4 (x,y,z,intensity). Returns count or <0. */
int orc_voxelgrid(const float* pts, int n, float leaf, float* out /* n*4 */);

/* K13: lidarOdometry.cpp LidarOdometry state machine on feature clouds. */
typedef struct orc_odom orc_odom;
orc_odom* orc_odom_create(const orc_params* reg_params, float ct_lambda);
void orc_odom_destroy(orc_odom* o);
void orc_odom_backend_odometry(orc_odom* o, double stamp); /* odometryHandler :440-446 (Q4 gate) */
/* cloudHandler :503-570 on one frame: corner/surface n*4 floats. pose6_out = LaserOdomPose (x,y,z,roll,pitch,yaw),
 * rot9/trans3 = Rotation / Translation after scanRegeistration. Returns 0 first frame, 1 gated (no registration),
 * 2 registered, <0 error. */
int orc_odom_cloud(orc_odom* o, double stamp, const float* corner, int n_corner, const float* surface, int n_surf,
                   float* pose6_out, double* rot9_out, double* trans3_out);

/* pcl::getTransformation / getTranslationAndEulerAngles (float) */
void orc_get_transformation(float x, float y, float z, float roll, float pitch, float yaw, float* T16);
void orc_get_translation_and_euler(const float* T16, float* xyzrpy6);
/* Eigen::Transform<float,3,Affine>::rotation() — the float polar factor of the linear part (lidarOdometry.cpp:130, 474, 548) */
void orc_affine3f_rotation(const float* T16, float* R9);

#ifdef __cplusplus
}
#endif
#endif
