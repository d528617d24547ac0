"""ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED by the reference itself.

Independent numpy statement of the per-frame odometry driver LidarOdometry (src/lidarOdometry.cpp: cloudHandler :503-570,
stateLinearPropagation :700-712, scanRegeistration :448-501, updateTransform :572-626), written from the reference source on
top of the registration twin (oracle/twin.py) and the float pose algebra of oracle/twin_front.py — a second opinion for
orc_odom_* in oracle/rolo_oracle_front.cpp. float32 where the reference uses Affine3f / float; numpy's float32 sin / cos and
its LAPACK 3x3 inverse differ from glibc / Eigen by ulps, so comparisons against the C++ oracle carry a ~1e-6 tolerance.
Conventions shared with the C++ oracle: cloudTimeLast starts at 0. Affine3f::rotation() is Eigen's polar factor of the linear part
(computeRotationScaling: float SVD, U V^T with the sign fix) — here through LAPACK's float SVD instead of Eigen's Jacobi sweeps.
"""
from __future__ import annotations

import numpy as np

from . import twin_front as tf
from .twin import Twin

F = np.float32


def mat4_mul_f(A, B):
    """Affine3f * Affine3f: float products summed k = 0..3 in order."""
    C = np.zeros((4, 4), F)
    for i in range(4):
        for j in range(4):
            s = F(0)
            for k in range(4):
                s = F(s + F(A[i, k] * B[k, j]))
            C[i, j] = s
    return C


def affine_inverse_f(T):
    """Eigen::Transform<float,3,Affine>::inverse(): inverse of the linear part, translation -inv * t."""
    out = np.eye(4, dtype=F)
    inv = np.linalg.inv(T[:3, :3].astype(F)).astype(F)
    out[:3, :3] = inv
    out[:3, 3] = -(inv @ T[:3, 3].astype(F)).astype(F)
    return out


def rotation_of_affine3f(T):
    """Eigen::Transform<float,3,Affine>::rotation(): JacobiSVD of the linear part in float, x = sign(det(U V^T)), U[:, 2] *= x, U V^T."""
    U, _, Vt = np.linalg.svd(np.asarray(T, F)[:3, :3])
    if np.linalg.det((U @ Vt).astype(np.float64)) < 0:
        U = U.copy(); U[:, 2] = -U[:, 2]
    return (U.astype(F) @ Vt.astype(F)).astype(F)


def transform_cloud_f(cloud4, T):
    """pcl::transformPointCloud with an Affine3f: x' = ((m00 x + m01 y) + m02 z) + m03, float; intensity kept."""
    T = np.asarray(T, F); c = np.asarray(cloud4, F); out = c.copy()
    x, y, z = c[:, 0], c[:, 1], c[:, 2]
    for r in range(3):
        out[:, r] = ((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]
    return out


class TwinOdom:
    def __init__(self, ct_lambda=0.3, polar_res=(0.175, 0.175, 2.0)):
        self.ct_lambda = F(ct_lambda); self.polar_res = polar_res
        self.first = True
        self.cloudTimeCur = 0.0; self.cloudTimeLast = 0.0       # Q3
        self.lastOdomTime = -1.0                                 # :417
        self.lastMappingInterval = 9999.0                        # :419
        self.lidarMappingAffine = np.eye(4, dtype=F)
        self.transformation_interpolated = np.eye(4, dtype=F)
        self.Rotation = np.eye(3); self.Translation = np.zeros(3); self.TranslationOld = np.zeros(3)   # :411-412
        self.LaserOdomPose = np.zeros(6, F)                      # :384
        self.featureOld = None

    def backend_odometry(self, stamp):                            # odometryHandler :440-446
        self.lastOdomTime = float(stamp)

    def _update_transform(self, featureLast):                     # :572-626 (pose part)
        trans = np.eye(4); trans[:3, :3] = self.Rotation; trans[:3, 3] = self.Translation
        step = trans.astype(F)
        pose = tf.get_transformation(*self.LaserOdomPose)
        moved = mat4_mul_f(pose, affine_inverse_f(step))
        self.lidarMappingAffine = step
        self.LaserOdomPose = tf.get_translation_and_euler(moved)
        self.featureOld = featureLast
        self.TranslationOld = self.Translation.copy()

    def cloud(self, stamp, corner, surface):
        self.cloudTimeCur = float(stamp)
        featureLast = np.concatenate([np.asarray(corner, F).reshape(-1, 4), np.asarray(surface, F).reshape(-1, 4)])   # :523
        if self.first:                                            # :525-532
            self.first = False
            self.featureOld = featureLast
            return 0
        if self.lastOdomTime == -1.0:                             # :536-540 (Q4)
            self._update_transform(featureLast)
            return 1
        latest = self.cloudTimeCur - self.cloudTimeLast           # :544
        ratio = latest / self.lastMappingInterval                 # stateLinearPropagation :700-712
        v = tf.get_translation_and_euler(self.lidarMappingAffine)
        v[3:] = 0
        v = (v * F(ratio)).astype(F)                              # float vector *= double: the scalar is cast to float
        self.transformation_interpolated = tf.get_transformation(*v)
        self.cloudTimeLast = self.cloudTimeCur
        self.lastMappingInterval = latest
        # scanRegeistration :448-501
        propagated = transform_cloud_f(self.featureOld, self.transformation_interpolated)
        tw = Twin(propagated, featureLast, voxel_type="polar", polar_res=self.polar_res)
        x0, _, _, _ = tw.align()
        step = x0.astype(F)                                       # getFinalTransformation() is a Matrix4f
        self.transformation_interpolated = mat4_mul_f(self.transformation_interpolated, step)
        self.Rotation = rotation_of_affine3f(self.transformation_interpolated).astype(np.float64)   # :474
        self.Translation = self.transformation_interpolated[:3, 3].astype(np.float64)
        reg_t, _, _ = tw.compute_translation(np.zeros(3), self.Translation, self.TranslationOld, 0.1, 0.1, self.ct_lambda)
        self.Translation = self.Translation + reg_t               # :500
        self._update_transform(featureLast)
        return 2
