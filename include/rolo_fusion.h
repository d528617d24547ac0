/* rolo_fusion.h — C ABI of the second half of the rolo_lidarOdometry node (part of librolo_hip.so; host code, no GPU needed):
 *
 *   rolo_eskf_*    rolo::eskf::PoseESEKF (include/rolo/eskf/eskf.hpp:39-358) — 18-dof constant-jerk error-state Kalman filter on
 *                  pos, rot (SO(3)), vel, omega, acc, alpha with a pose measurement, on the iterated ESKF of the IKFoM toolkit
 *                  (include/rolo/eskf/IKFoM_toolkit/esekfom/esekfom.hpp: predict :275-403, update_iterated :406-703), as written
 *                  (incl. scalar_type(1/2) == 0 at :359 — the SO(3) block of F_x1 is the identity);
 *   rolo_fusion_*  TransformFusion (src/lidarOdometry.cpp:47-323): mappingOdometryHandler :109-117, lidarOdometryHandler :119-125,
 *                  fusionTimerHandler :138-241 (20 Hz odomTopic, speed, 1 s path), predictTimerHandler :243-322 (30 Hz future path).
 *
 * Quaternions are x, y, z, w (the message order); times are seconds (header.stamp.toSec() / ros::Time::now().toSec()).
 * Functions return 0 / 1 where the reference returns bool, negative ROLO_E* on misuse.
 */
#ifndef ROLO_FUSION_H
#define ROLO_FUSION_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rolo_eskf_options {   /* PoseESEKF::Options, eskf.hpp:55-69 */
  double max_dt, q_linear_jerk_std, q_angular_jerk_std, r_position_std, r_rotation_std;
  double init_position_std, init_rotation_std, init_velocity_std, init_angular_velocity_std, init_acceleration_std, init_angular_acceleration_std;
  int maximum_iteration;
  double convergence_limit;
} rolo_eskf_options;
void rolo_eskf_default_options(rolo_eskf_options* o);

typedef struct rolo_eskf rolo_eskf;
int rolo_eskf_create(const rolo_eskf_options* options_or_null, rolo_eskf** out);
void rolo_eskf_destroy(rolo_eskf* f);
int rolo_eskf_copy(const rolo_eskf* src, rolo_eskf* dst);                 /* pose_preview = pose_regulator (lidarOdometry.cpp:179) */
void rolo_eskf_reset(rolo_eskf* f);                                       /* :90-95 */
int rolo_eskf_initialized(const rolo_eskf* f);                            /* :82-84 */
double rolo_eskf_last_time(const rolo_eskf* f);                           /* :86-88 */
/* processMeasurement :108-147; R36 = 6x6 row-major measurement noise or NULL for defaultMeasurementNoise() :189-196 */
int rolo_eskf_process_measurement(rolo_eskf* f, double stamp, const double position3[3], const double orientation_xyzw[4], const double* R36);
int rolo_eskf_state_predict(rolo_eskf* f, double stamp);                  /* :149-171 */
/* position / orientation (normalised, x y z w) / velocity / angularVelocity :173-187; acc / alpha = the other two state blocks. NULLs skipped */
void rolo_eskf_get_state(const rolo_eskf* f, double position3[3], double orientation_xyzw[4], double velocity3[3], double omega3[3], double acc3[3], double alpha3[3]);
void rolo_eskf_get_covariance(const rolo_eskf* f, double P324[324]);      /* 18 x 18 row-major: pos rot vel omega acc alpha */
/* statePropagate :213-246: poses7 = cap x (x y z qx qy qz qw); returns the number of poses (may exceed cap: only cap are written) */
int rolo_eskf_state_propagate(const rolo_eskf* f, double dt, double dis, double* poses7, int cap);

typedef struct rolo_fusion rolo_fusion;
int rolo_fusion_create(const rolo_eskf_options* options_or_null, rolo_fusion** out);
void rolo_fusion_destroy(rolo_fusion* f);
/* mappingOdometryHandler / lidarOdometryHandler: the pose of the nav_msgs/Odometry and its header stamp */
int rolo_fusion_mapping_odometry(rolo_fusion* f, double stamp, const double position3[3], const double orientation_xyzw[4]);
int rolo_fusion_lidar_odometry(rolo_fusion* f, double stamp, const double position3[3], const double orientation_xyzw[4]);
typedef struct rolo_fusion_odometry {   /* what fusionTimerHandler publishes on odomTopic / odomTopic + "/speed" */
  double position[3], orientation[4] /* x y z w */, velocity[3], speed;
  int path_appended;   /* a pose went onto rolo/lidar_odometry/path (at most every 0.05 s, 1 s kept) */
  int path_length;
} rolo_fusion_odometry;
/* fusionTimerHandler at time `now`: 1 = `out` is to be published, 0 = the handler returned early */
int rolo_fusion_timer(rolo_fusion* f, double now, rolo_fusion_odometry* out);
typedef struct rolo_future_point {      /* autoware_rviz_msgs/PathPoint as predictTimerHandler fills it */
  double position[3], orientation[4], longitudinal_velocity_mps, lateral_velocity_mps, heading_rate_rps;
  int is_final;
} rolo_future_point;
/* predictTimerHandler: returns the number of future points (0 = nothing published); at most cap are written */
int rolo_fusion_predict_timer(rolo_fusion* f, rolo_future_point* points, int cap);
rolo_eskf* rolo_fusion_filter(rolo_fusion* f);   /* pose_regulator (owned by the fusion object) */

#ifdef __cplusplus
}
#endif
#endif /* ROLO_FUSION_H */
