"""GPU: the whole hot path end to end — raw PointXYZIRT frames -> projection -> features -> LidarOdometry — against
the oracle running the same chain on the CPU. Poses must agree to <= 1e-5 rad / <= 1e-4 m (BASELINE north_star)."""
import numpy as np
import pytest

from oracle import pyorc
from rolo_amd import synth
from rolo_amd.frontend import FrontEnd, front_params
from rolo_amd.odometry import LidarOdometry

pytestmark = pytest.mark.gpu


def trajectory(n):
    """Sensor poses: gentle arc, ~0.3 m and ~2 deg yaw per frame."""
    poses = []
    R = np.eye(3); t = np.zeros(3)
    for k in range(n):
        poses.append((R.copy(), t.copy()))
        dR = synth.rpy_to_R(np.deg2rad(0.3), np.deg2rad(-0.2), np.deg2rad(2.0))
        t = t + R @ np.array([0.3, 0.02 * k, 0.0])
        R = R @ dR
    return poses


@pytest.mark.parametrize("sensor,cfg", [("vlp16", dict(n_scan=16, horizon_scan=1800)), ("os1-64", dict(n_scan=64, horizon_scan=1024))])
def test_pipeline_matches_oracle(sensor, cfg):
    poses = trajectory(5)
    fo = pyorc.front_params(**cfg)
    fg = front_params(**cfg)
    oo = pyorc.Odom(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0)), 0.3)
    og = LidarOdometry(0, 0.3)
    fe = FrontEnd(og.reg, fg)
    statuses = []
    for k, (R, t) in enumerate(poses):
        fr = synth.make_frame(sensor, R, t, synth.SEED + k)
        po = pyorc.project(fo, fr.xyz, fr.ring)
        eo = pyorc.extract_features(fo, po)
        pg = fe.project(fr.xyz, fr.ring)
        eg = fe.extract(pg["n"])
        assert np.array_equal(eg["corner"], eo["corner"]) and np.array_equal(eg["surface"], eo["surface"])
        stamp = 100.0 + 0.1 * k
        if k == 2:  # the back end publishes its first odometry after the second frame
            oo.backend_odometry(stamp - 0.05); og.odometryHandler(stamp - 0.05)
        rco, pose_o, R_o, t_o = oo.cloud(stamp, eo["corner"], eo["surface"])
        rcg, pose_g, R_g, t_g = og.cloudHandler(stamp, eg["corner"], eg["surface"])
        statuses.append(rcg)
        assert rcg == rco
        assert np.abs(pose_g[:3] - pose_o[:3]).max() <= 1e-4 and np.abs(pose_g[3:] - pose_o[3:]).max() <= 1e-5
        assert np.abs(R_g - R_o).max() <= 1e-5 and np.abs(t_g - t_o).max() <= 1e-4
        assert np.abs(pose_g - pose_o).max() < 2e-6  # float32 pose chain: a few ulps
    assert statuses == [0, 1, 2, 2, 2]
    # and the estimate is sane: the last step's translation is within a few cm of the true sensor motion
    Rk, tk = poses[-2]; Rn, tn = poses[-1]
    true_step = -(Rn.T @ (tn - tk))
    assert np.linalg.norm(t_g - true_step) < 0.1
