"""TEST INFRASTRUCTURE — independent numpy/scipy statement of rolo::eskf::PoseESEKF (reference include/rolo/eskf/eskf.hpp:39-358 on the
iterated error-state Kalman filter of the IKFoM toolkit, include/rolo/eskf/IKFoM_toolkit/esekfom/esekfom.hpp: predict :275-403,
update_iterated :406-703) and of TransformFusion's two timers (src/lidarOdometry.cpp:47-323). Written from the reference sources on library
routines (numpy.linalg, scipy Rotation) instead of the hand-rolled loops of rolo_amd/csrc/fusion.hip; the tests hold the C ABI
(include/rolo_fusion.h) to it. Quaternions are x, y, z, w.

State order (18 dof): pos 0:3, rot 3:6 (SO(3)), vel 6:9, omega 9:12, acc 12:15, alpha 15:18.
As-written quirk kept: esekfom.hpp:359 passes scalar_type(1/2) == 0 as the scale of the SO(3) block of F_x1, i.e. that block is the identity.
"""
import copy

import numpy as np
from scipy.spatial.transform import Rotation

TOL = 1e-11


def hat(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], float)


def A_matrix(v):   # mtkmath.hpp:235-247
    sq = float(v @ v); n = np.sqrt(sq)
    if n < TOL:
        return np.eye(3)
    H = hat(v)
    return np.eye(3) + (1 - np.cos(n)) / sq * H + (1 - np.sin(n) / n) / sq * (H @ H)


def qmul(a, b):   # x y z w
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def qconj(a):
    return np.array([-a[0], -a[1], -a[2], a[3]])


def so3_exp(v, scale=1.0):   # SO3::exp: the rotation by scale * v as a quaternion (exact sin / cos; the reference switches to a Taylor series
    s = scale / 2.0          # below |x|^2 < eps^(1/4), which agrees to rounding)
    ang = s * np.linalg.norm(v)
    sinc = np.sin(ang) / ang if ang > 1e-8 else 1.0 - ang * ang / 6.0
    return np.concatenate([sinc * s * np.asarray(v, float), [np.cos(ang)]])


def so3_log(q):   # SO3::log with plus_minus_periodicity
    nv = np.linalg.norm(q[:3])
    if nv < TOL:
        nv = TOL
    return 2.0 / nv * np.arctan(nv / q[3]) * q[:3]


class Options:
    def __init__(self, **kw):
        self.max_dt = 1.0; self.q_linear_jerk_std = 0.5; self.q_angular_jerk_std = 0.5; self.r_position_std = 0.20; self.r_rotation_std = 0.10
        self.init_position_std = 0.05; self.init_rotation_std = 0.05; self.init_velocity_std = 5.0; self.init_angular_velocity_std = 2.0
        self.init_acceleration_std = 5.0; self.init_angular_acceleration_std = 2.0; self.maximum_iteration = 3; self.convergence_limit = 1e-4
        for k, v in kw.items():
            setattr(self, k, v)


class PoseESEKF:
    def __init__(self, options=None):
        self.o = options or Options()
        self.Q = np.diag([self.o.q_linear_jerk_std ** 2] * 3 + [self.o.q_angular_jerk_std ** 2] * 3)
        self.x = np.zeros(18); self.q = np.array([0, 0, 0, 1.0])   # x[3:6] unused: the rotation lives in q
        self.reset()

    def _P0(self):
        o = self.o
        sd = [o.init_position_std, o.init_rotation_std, o.init_velocity_std, o.init_angular_velocity_std, o.init_acceleration_std, o.init_angular_acceleration_std]
        return np.diag(np.repeat(np.square(sd), 3))

    def reset(self):
        self.initialized = False; self.last_time = 0.0; self.P = self._P0()

    @staticmethod
    def _normq(q):
        q = np.asarray(q, float)
        if not np.all(np.isfinite(q)) or np.linalg.norm(q) < 1e-12:
            return np.array([0, 0, 0, 1.0])
        return q / np.linalg.norm(q)

    def initialize(self, stamp, p, q):
        self.x = np.zeros(18); self.x[:3] = p; self.q = self._normq(q); self.P = self._P0(); self.initialized = True; self.last_time = stamp

    def _f(self, x, dt):
        f = np.zeros(18)
        f[0:3] = x[6:9] + 0.5 * dt * x[12:15]; f[3:6] = x[9:12] + 0.5 * dt * x[15:18]; f[6:9] = x[12:15]; f[9:12] = x[15:18]
        return f

    def _plus(self, x, q, d, scale):
        x = x + scale * d; x[3:6] = 0
        return x, qmul(q, so3_exp(d[3:6], scale))

    def predict(self, dt):
        f = self._f(self.x, dt)
        Fx = np.zeros((18, 18)); I3 = np.eye(3)
        Fx[0:3, 6:9] = I3; Fx[0:3, 12:15] = 0.5 * dt * I3; Fx[3:6, 9:12] = I3; Fx[3:6, 15:18] = 0.5 * dt * I3; Fx[6:9, 12:15] = I3; Fx[9:12, 15:18] = I3
        Fw = np.zeros((18, 6)); Fw[12:15, 0:3] = I3; Fw[15:18, 3:6] = I3
        self.x, self.q = self._plus(self.x, self.q, f, dt)
        A = A_matrix(-f[3:6] * dt)
        Fx[3:6, :] = A @ Fx[3:6, :]; Fw[3:6, :] = A @ Fw[3:6, :]
        F = np.eye(18) + Fx * dt   # SO(3) block of F_x1 is Identity as written
        G = dt * Fw
        self.P = F @ self.P @ F.T + G @ self.Q @ G.T

    def _minus(self, x, q, xp, qp):
        d = x - xp
        d[3:6] = so3_log(qmul(qconj(qp), q))
        return d

    def update_iterated(self, zp, zq, R):
        xp, qp, Pp = self.x.copy(), self.q.copy(), self.P.copy()
        H = np.zeros((6, 18)); H[:3, :3] = np.eye(3); H[3:, 3:6] = np.eye(3)
        t = 0
        for it in range(self.o.maximum_iteration):
            dx = self._minus(self.x, self.q, xp, qp)
            J = np.eye(18); J[3:6, 3:6] = A_matrix(dx[3:6]).T
            dx_new = J @ dx
            P = J @ Pp @ J.T
            K = P @ H.T @ np.linalg.inv(H @ P @ H.T + R)
            innov = np.concatenate([zp - self.x[:3], so3_log(qmul(qconj(self.q), zq))])
            dxu = K @ innov + (K @ H - np.eye(18)) @ dx_new
            self.x, self.q = self._plus(self.x, self.q, dxu, 1.0)
            if np.all(np.abs(dxu) <= self.o.convergence_limit):
                t += 1
            if t > 1 or it == self.o.maximum_iteration - 1:
                J2 = np.eye(18); J2[3:6, 3:6] = A_matrix(dxu[3:6]).T
                Lm = J2 @ P @ J2.T          # rows then columns of the SO(3) block
                K2 = J2 @ K
                P2 = P @ J2.T               # P gets the column transform only (esekfom.hpp:640-643)
                self.P = Lm - K2 @ H @ P2
                return

    def default_R(self):
        return np.diag([self.o.r_position_std ** 2] * 3 + [self.o.r_rotation_std ** 2] * 3)

    def process_measurement(self, stamp, p, q, R=None):
        R = self.default_R() if R is None else np.array(R, float).reshape(6, 6)
        if not self.initialized:
            self.initialize(stamp, p, q); return True
        dt = stamp - self.last_time
        if dt <= 0.0 or not np.isfinite(dt):
            return False
        if dt > self.o.max_dt:
            self.initialize(stamp, p, q); return True
        self.predict(dt)
        for i in range(6):
            if not np.isfinite(R[i, i]) or R[i, i] < 1e-12:
                R[i, i] = 1e-12
        self.update_iterated(np.asarray(p, float), self._normq(q), R)
        self.last_time = stamp
        return True

    def state_predict(self, stamp):
        if not self.initialized:
            return False
        dt = stamp - self.last_time
        if dt <= 0.0 or not np.isfinite(dt) or dt > self.o.max_dt:
            return False
        self.predict(dt); self.last_time = stamp
        return True

    def orientation(self):
        return self.q / np.linalg.norm(self.q)

    def state_propagate(self, dt, dis):
        out = []
        if not self.initialized or dt <= 0 or dis <= 0:
            return out
        x, q = self.x.copy(), self.q.copy(); last = x[:3].copy(); prop = 0.0
        while prop < dis:
            x, q = self._plus(x, q, self._f(x, dt), dt)
            step = np.linalg.norm(x[:3] - last)
            if not np.isfinite(step) or step < 1e-12:
                break
            prop += step; last = x[:3].copy()
            out.append(np.concatenate([x[:3], q / np.linalg.norm(q)]))
        return out


def get_rpy(q):   # tf::Matrix3x3(q).getRPY away from gimbal lock = extrinsic x, y, z Euler angles
    return Rotation.from_quat(q).as_euler("xyz")


def odom2affine(p, q):
    """odom2affine (lidarOdometry.cpp:34-45): float Affine3f of pcl::getTransformation(x, y, z, roll, pitch, yaw)"""
    r, pi, y = np.float32(get_rpy(q))
    A, B, C, D, E, F = np.cos(y), np.sin(y), np.cos(pi), np.sin(pi), np.cos(r), np.sin(r)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[A * C, A * D * F - B * E, B * F + A * D * E], [B * C, A * E + B * D * F, B * D * E - A * F], [-D, C * F, C * E]], np.float32)
    T[:3, 3] = np.float32(p)
    return T


class TransformFusion:
    def __init__(self, options=None):
        self.kf = PoseESEKF(options)
        self.mapping = np.eye(4, dtype=np.float32); self.mapping_time = -1.0; self.last_processed = -1.0; self.last_path = -1.0
        self.queue = []; self.path = []

    def mapping_odometry(self, stamp, p, q):
        self.mapping = odom2affine(p, q); self.mapping_time = stamp

    def lidar_odometry(self, stamp, p, q):
        self.queue.append((stamp, np.asarray(p, float), np.asarray(q, float)))

    @staticmethod
    def _pose(T):
        # affine.rotation() = the polar factor of the float linear part (Eigen: a float Jacobi SVD; scipy: an SVD-based orthonormalisation
        # in double — the same matrix to float rounding); Quaterniond(...).normalize(). Eigen's sign convention is fixed by comparing rotations
        q = Rotation.from_matrix(T[:3, :3].astype(np.float64)).as_quat()
        return T[:3, 3].astype(np.float64), q

    def timer(self, now):
        if self.mapping_time == -1:
            return None
        while self.queue and self.queue[0][0] <= self.mapping_time:
            self.queue.pop(0)
        if not self.queue:
            return None
        front = odom2affine(self.queue[0][1], self.queue[0][2])
        if self.queue[-1][0] > self.last_processed:
            st, p, q = self.queue[-1]
            mp, mq = self._pose(odom2affine(p, q))
            if self.kf.process_measurement(st, mp, mq):
                self.last_processed = st
        if not self.kf.initialized:
            return None
        pv = copy.deepcopy(self.kf)
        pv.state_predict(now)
        back = np.eye(4, dtype=np.float32)
        back[:3, :3] = Rotation.from_quat(pv.orientation()).as_matrix().astype(np.float32); back[:3, 3] = pv.x[:3].astype(np.float32)
        incre = (np.linalg.inv(front.astype(np.float64)) @ back.astype(np.float64)).astype(np.float32)
        last = (self.mapping.astype(np.float64) @ incre.astype(np.float64)).astype(np.float32)
        pos, q = self._pose(last)
        appended = False
        if now - self.last_path > 0.05:
            self.last_path = now; self.path.append(now)
            while self.path and self.path[0] < now - 1.0:
                self.path.pop(0)
            appended = True
        return dict(position=pos, orientation=q, velocity=pv.x[6:9].copy(), speed=float(np.linalg.norm(pv.x[6:9])), path_appended=appended, path_length=len(self.path))

    def predict_timer(self):
        if not self.kf.initialized:
            return []
        poses = self.kf.state_propagate(0.2, 8.0)
        Rc = Rotation.from_quat(self.kf.orientation()).as_matrix()
        lv = Rc.T @ self.kf.x[6:9]
        out = []
        for i, p in enumerate(poses):
            Rf = Rotation.from_quat(p[3:] / np.linalg.norm(p[3:])).as_matrix()
            tl = Rc.T @ (p[:3] - self.kf.x[:3])
            out.append(dict(position=np.array([tl[0], tl[1], 0.0]), R=Rc.T @ Rf, longitudinal=lv[0], lateral=lv[1], heading_rate=self.kf.x[11], is_final=i + 1 == len(poses)))
        return out
