"""CPU (host code, no GPU needed): PoseESEKF and TransformFusion behind include/rolo_fusion.h against
  * the fixture extracted from the reference's own resource/test_odom.bag (2729 front-end odometry messages; tests/golden/make_golden_eskf.py)
    with the independent numpy twin's filtered output,
  * the live twin (oracle/twin_eskf.py) on synthetic sequences incl. rejected / re-initialising stamps,
  * analytic behaviour: a constant-velocity track is recovered, a static pose is held with a steady-state covariance below the measurement noise."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import twin_eskf
from rolo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(dp)


class Eskf:
    def __init__(self, h=None, **opt):
        self.L = _lib.lib()
        if h is not None:
            self.h, self.own = h, False
            return
        o = _lib.EskfOptions(); self.L.rolo_eskf_default_options(C.byref(o))
        for k, v in opt.items():
            setattr(o, k, v)
        self.h = C.c_void_p(); self.own = True
        assert self.L.rolo_eskf_create(C.byref(o), C.byref(self.h)) == 0

    def __del__(self):
        if getattr(self, "own", False) and self.h:
            self.L.rolo_eskf_destroy(self.h); self.h = None

    def process(self, stamp, p, q, R=None):
        p = np.ascontiguousarray(p, np.float64); q = np.ascontiguousarray(q, np.float64)
        Rm = np.ascontiguousarray(R, np.float64) if R is not None else None
        return self.L.rolo_eskf_process_measurement(self.h, stamp, _d(p), _d(q), _d(Rm) if Rm is not None else None)

    def state(self):
        out = [np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)]
        self.L.rolo_eskf_get_state(self.h, *[_d(a) for a in out])
        return np.concatenate(out)

    def P(self):
        P = np.zeros((18, 18)); self.L.rolo_eskf_get_covariance(self.h, _d(P)); return P


def twin_state(kf):
    return np.concatenate([kf.x[:3], kf.orientation(), kf.x[6:18]])


def test_reference_bag_fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "test_odom_bag.npz"))
    stamp = z["sec"].astype(np.float64) + 1e-9 * z["nsec"].astype(np.float64)
    pose = z["pose"]
    assert pose.shape == (2729, 7)
    f = Eskf()
    got = []
    for k in range(pose.shape[0]):
        assert f.process(stamp[k], pose[k, :3], pose[k, 3:]) == int(z["accepted"][k])
        if k % 8 == 0:
            got.append(f.state())
    got = np.array(got); want = z["filtered_every8"]
    # quaternion sign is a convention: compare up to sign
    sgn = np.sign(np.sum(got[:, 3:7] * want[:, 3:7], axis=1))[:, None]
    got[:, 3:7] *= sgn
    scale = np.maximum(np.abs(want).max(axis=0), 1.0)
    assert (np.abs(got - want) / scale).max() < 1e-9
    last = f.state(); wl = z["filtered_last"].copy()
    if np.dot(last[3:7], wl[3:7]) < 0:
        wl[3:7] *= -1
    assert np.abs(last - wl).max() < 1e-8
    assert np.abs(f.P() - z["final_P"]).max() <= 1e-9 * np.abs(z["final_P"]).max()
    # the filter follows the recorded track: filtered position stays within a few cm of the measurement at 10 Hz
    assert np.linalg.norm(last[:3] - pose[-1, :3]) < 0.2


def _track(n, seed=3):
    g = np.random.default_rng(seed)
    t = 100.0 + np.cumsum(g.uniform(0.05, 0.15, n))
    yaw = 0.15 * (t - t[0]); pos = np.c_[2.0 * (t - t[0]), 0.3 * np.sin(0.5 * (t - t[0])), 0.02 * (t - t[0])]
    q = Rotation.from_euler("xyz", np.c_[0.01 * np.sin(t), 0.02 * np.cos(t), yaw]).as_quat()
    pos += g.normal(0, 0.01, pos.shape)
    return t, pos, q


def test_matches_live_twin_incl_rejected_and_reinitialising_stamps():
    t, pos, q = _track(200)
    t[50] = t[49] - 0.01          # time going backwards: rejected
    t[120:] += 5.0                # gap > max_dt: re-initialise
    R = np.diag([0.04, 0.05, 0.06, 0.01, 0.02, 0.03])
    f = Eskf(); kf = twin_eskf.PoseESEKF()
    for k in range(len(t)):
        Rk = R if k % 3 == 0 else None
        a = f.process(t[k], pos[k], q[k], Rk); b = kf.process_measurement(t[k], pos[k], q[k], Rk)
        assert a == int(b)
        s, w = f.state(), twin_state(kf)
        if np.dot(s[3:7], w[3:7]) < 0:
            w[3:7] *= -1
        assert np.abs(s - w).max() < 1e-9 * max(1.0, np.abs(w).max())
        assert np.abs(f.P() - kf.P).max() <= 1e-9 * np.abs(kf.P).max()
    assert f.L.rolo_eskf_last_time(f.h) == kf.last_time
    # statePredict + statePropagate
    assert f.L.rolo_eskf_state_predict(f.h, t[-1] + 0.03) == 1 and kf.state_predict(t[-1] + 0.03)
    assert f.L.rolo_eskf_state_predict(f.h, t[-1] - 1.0) == 0 and not kf.state_predict(t[-1] - 1.0)
    want = kf.state_propagate(0.2, 8.0)
    buf = np.zeros((len(want) + 4, 7))
    n = f.L.rolo_eskf_state_propagate(f.h, 0.2, 8.0, _d(buf), buf.shape[0])
    assert n == len(want) and np.abs(buf[:n] - np.array(want)).max() < 1e-9
    assert f.L.rolo_eskf_state_propagate(f.h, -1.0, 8.0, None, 0) == 0


def test_constant_velocity_is_recovered_and_static_pose_contracts():
    f = Eskf()
    v = np.array([1.5, -0.4, 0.1])
    for k in range(300):
        f.process(10.0 + 0.1 * k, v * 0.1 * k, [0, 0, 0, 1])
    s = f.state()
    assert np.abs(s[7:10] - v).max() < 1e-3 and np.abs(s[13:16]).max() < 1e-2   # velocity found, no acceleration
    g = Eskf()
    g.process(1.0, [1, 2, 3], [0, 0, 0, 1])
    for k in range(1, 300):
        g.process(1.0 + 0.1 * k, [1, 2, 3], [0, 0, 0, 1])
        if k == 250:
            P250 = g.P()
    # a static pose: the estimate stays put, the covariance settles at a steady state below the measurement noise (0.2 m, 0.1 rad)
    assert np.abs(g.state()[:3] - [1, 2, 3]).max() < 1e-6 and np.abs(g.state()[7:10]).max() < 1e-6
    assert np.abs(g.P() - P250).max() < 1e-6 and np.trace(g.P()[:3, :3]) < 3 * 0.2 ** 2 and np.trace(g.P()[3:6, 3:6]) < 3 * 0.1 ** 2
    P = g.P(); assert np.abs(P - P.T).max() < 1e-9 and np.linalg.eigvalsh(0.5 * (P + P.T)).min() > -1e-9


def test_transform_fusion_timers_match_twin():
    L = _lib.lib()
    h = C.c_void_p(); assert L.rolo_fusion_create(None, C.byref(h)) == 0
    tw = twin_eskf.TransformFusion()
    t, pos, q = _track(80, seed=5)
    out = _lib.FusionOdometry()
    assert L.rolo_fusion_timer(h, t[0], C.byref(out)) == 0 and tw.timer(t[0]) is None     # no back-end odometry yet
    mp = np.array([0.5, -0.2, 0.1]); mq = Rotation.from_euler("xyz", [0.01, -0.02, 0.3]).as_quat()
    n_pub = 0
    for k in range(len(t)):
        pk = np.ascontiguousarray(pos[k]); qk = np.ascontiguousarray(q[k])
        L.rolo_fusion_lidar_odometry(h, t[k], _d(pk), _d(qk)); tw.lidar_odometry(t[k], pk, qk)
        if k == 10 or k == 40:   # the back end publishes at the stamp of an earlier scan
            L.rolo_fusion_mapping_odometry(h, t[k - 2], _d(mp), _d(mq)); tw.mapping_odometry(t[k - 2], mp, mq)
        for now in (t[k] + 0.02, t[k] + 0.045):   # 20 Hz-ish timer ticks between scans
            r = L.rolo_fusion_timer(h, now, C.byref(out)); w = tw.timer(now)
            assert (r == 1) == (w is not None)
            if w is None:
                continue
            n_pub += 1
            assert np.abs(np.array(out.position) - w["position"]).max() < 2e-5           # float Affine3f chain
            Ra = Rotation.from_quat(np.array(out.orientation)).as_matrix(); Rb = Rotation.from_quat(w["orientation"]).as_matrix()
            assert np.abs(Ra - Rb).max() < 2e-6
            assert np.abs(np.array(out.velocity) - w["velocity"]).max() < 1e-5 and abs(out.speed - w["speed"]) < 1e-5
            assert bool(out.path_appended) == w["path_appended"] and out.path_length == w["path_length"]
        pts = (_lib.FuturePoint * 512)()
        npt = L.rolo_fusion_predict_timer(h, pts, 512); wp = tw.predict_timer()
        assert npt == len(wp)
        for i in range(npt):
            # the measurements reach the filter through float Affine3f matrices (odom2affine) and Affine3f::rotation(): the library restates
            # Eigen's float Jacobi SVD (polar factor), the twin orthonormalises in double: agreement at float rounding, a few 1e-7
            assert np.abs(np.array(pts[i].position) - wp[i]["position"]).max() < 1e-5 and pts[i].position[2] == 0.0
            assert np.abs(Rotation.from_quat(np.array(pts[i].orientation)).as_matrix() - wp[i]["R"]).max() < 5e-6   # (a 1e-7 difference of a measured rotation shows up amplified in the propagated future poses)
            assert abs(pts[i].longitudinal_velocity_mps - wp[i]["longitudinal"]) < 1e-5 and abs(pts[i].heading_rate_rps - wp[i]["heading_rate"]) < 1e-5
            assert bool(pts[i].is_final) == wp[i]["is_final"]
    assert n_pub > 100
    L.rolo_fusion_destroy(h)
