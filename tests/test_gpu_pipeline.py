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


@pytest.mark.parametrize("sensor,cfg", [("vlp16", dict(n_scan=16, horizon_scan=1800)), ("os1-64", dict(n_scan=64, horizon_scan=1024))])
def test_fused_frame_equals_staged_nodes(sensor, cfg):
    """rolo_odom_frame (raw frame -> pose with device-resident hand-over) gives exactly what the three staged node cores
    give — from host buffers and from device pointers — and with ROLO_ODOM_REUSE_COVARIANCES the poses stay inside the
    north_star tolerance of the oracle chain (covariances of the propagated source taken over from the last target)."""
    import torch
    poses = trajectory(6)
    fo = pyorc.front_params(**cfg)
    fg = front_params(**cfg)
    oo = pyorc.Odom(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0)), 0.3)
    staged = LidarOdometry(0, 0.3); fe = FrontEnd(staged.reg, fg)
    fused_h = LidarOdometry(0, 0.3)
    fused_d = LidarOdometry(0, 0.3)
    fused_r = LidarOdometry(0, 0.3); fused_r.setOption(LidarOdometry.REUSE_COVARIANCES, 1)
    fused_p = LidarOdometry(0, 0.3)   # submit / collect: K1-K4 of frame k+1 overlap the registration of frame k
    frames = [synth.make_frame(sensor, R, t, synth.SEED + k) for k, (R, t) in enumerate(poses)]
    fused_p.submit(fg, 100.0, frames[0].xyz, frames[0].ring)
    for k, (R, t) in enumerate(poses):
        fr = frames[k]
        stamp = 100.0 + 0.1 * k
        if k == 2:
            for o in (staged, fused_h, fused_d, fused_r, fused_p):
                o.odometryHandler(stamp - 0.05)
            oo.backend_odometry(stamp - 0.05)
        eo = pyorc.extract_features(fo, pyorc.project(fo, fr.xyz, fr.ring))
        rco, pose_o, R_o, t_o = oo.cloud(stamp, eo["corner"], eo["surface"])
        pg = fe.project(fr.xyz, fr.ring); eg = fe.extract(pg["n"])
        rcs, pose_s, R_s, t_s = staged.cloudHandler(stamp, eg["corner"], eg["surface"])
        rch, pose_h, R_h, t_h, cnt = fused_h.frame(fg, stamp, fr.xyz, fr.ring)
        assert cnt == (pg["n"], eg["corner"].shape[0], eg["surface"].shape[0])
        fc, fs = fused_h.features()   # what the outgoing CloudInfoStamp carries
        assert np.array_equal(fc, eg["corner"]) and np.array_equal(fs, eg["surface"])
        d_xyz = torch.from_numpy(np.ascontiguousarray(fr.xyz, np.float32)).cuda()
        d_ring = torch.from_numpy(np.ascontiguousarray(fr.ring, np.uint16).view(np.int16)).cuda()
        torch.cuda.synchronize()
        rcd, pose_d, R_d, t_d, cnt_d = fused_d.frame(fg, stamp, d_xyz.data_ptr(), d_ring.data_ptr(), n_raw=fr.xyz.shape[0], stride=fr.xyz.shape[1])
        assert rcs == rch == rcd == rco and cnt_d == cnt
        for (p_, R_, t_) in ((pose_h, R_h, t_h), (pose_d, R_d, t_d)):
            # same kernels on the same bytes; only the fp64 atomics of the voxel sums may reorder
            assert np.abs(p_ - pose_s).max() < 1e-6 and np.abs(R_ - R_s).max() < 1e-9 and np.abs(t_ - t_s).max() < 1e-9
        if k + 1 < len(frames):
            fused_p.submit(fg, stamp + 0.1, frames[k + 1].xyz, frames[k + 1].ring)
        rcp, pose_p, R_p, t_p, cnt_p = fused_p.collect()
        assert rcp == rch and cnt_p == cnt
        assert np.abs(pose_p - pose_h).max() < 1e-6 and np.abs(R_p - R_h).max() < 1e-9 and np.abs(t_p - t_h).max() < 1e-9
        rcr, pose_r, R_r, t_r, _ = fused_r.frame(fg, stamp, fr.xyz, fr.ring)
        assert rcr == rco
        assert np.abs(pose_r[:3] - pose_o[:3]).max() <= 1e-4 and np.abs(pose_r[3:] - pose_o[3:]).max() <= 1e-5
        assert np.abs(R_r - R_o).max() <= 1e-5 and np.abs(t_r - t_o).max() <= 1e-4


def test_submit_collect_queue_rules():
    from rolo_amd._lib import RoloError
    cfg = dict(n_scan=16, horizon_scan=1800)
    fg = front_params(**cfg)
    od = LidarOdometry(0, 0.3)
    with pytest.raises(RoloError) as ei:
        od.collect()                                   # nothing submitted
    assert ei.value.code == -5
    frames = [synth.make_frame("vlp16", R, t, synth.SEED + k) for k, (R, t) in enumerate(trajectory(3))]
    od.submit(fg, 100.0, frames[0].xyz, frames[0].ring)
    od.submit(fg, 100.1, frames[1].xyz, frames[1].ring)
    with pytest.raises(RoloError) as ei:
        od.submit(fg, 100.2, frames[2].xyz, frames[2].ring)   # at most two frames in flight
    assert ei.value.code == -5
    with pytest.raises(RoloError) as ei:
        od.frame(fg, 100.2, frames[2].xyz, frames[2].ring)    # frame() needs an empty queue
    assert ei.value.code == -5
    assert od.collect()[0] == 0 and od.collect()[0] == 1      # first frame stored, second gated
    od.odometryHandler(100.15)
    assert od.frame(fg, 100.2, frames[2].xyz, frames[2].ring)[0] == 2
    # an empty frame is an error code of the registration, not a crash
    empty = np.zeros((0, 3), np.float32)
    with pytest.raises(RoloError):
        od.frame(fg, 100.3, np.full((64, 3), 0.5, np.float32), np.zeros(64, np.uint16))   # everything below lidarMinRange
    del empty


def test_cpp_node_cores_end_to_end(tmp_path):
    """include/rolo_nodes_hip.hpp (rolo::ImageProjection / FeatureExtraction / LidarOdometry) in a C++-only process:
    the staged chain and the fused submit/collect path give the oracle's poses on a raw frame sequence."""
    import os, struct, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "nodes_demo")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "nodes_demo.cpp"), "-o", exe,
           "-L", os.path.join(root, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(root, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    cfg = dict(n_scan=16, horizon_scan=1800)
    poses = trajectory(5)
    frames = [synth.make_frame("vlp16", R, t, synth.SEED + k) for k, (R, t) in enumerate(poses)]
    with open(tmp_path / "frames.bin", "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for fr in frames:
            f.write(struct.pack("<i", fr.xyz.shape[0]))
            f.write(np.ascontiguousarray(fr.xyz, np.float32).tobytes()); f.write(np.ascontiguousarray(fr.ring, np.uint16).tobytes())
    r = subprocess.run([exe, str(tmp_path / "frames.bin"), "16", "1800"], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "error -5"      # collect() with nothing submitted -> ROLO_ESTATE as rolo::Error
    fo = pyorc.front_params(**cfg)
    oo = pyorc.Odom(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0)), 0.3)
    for k, fr in enumerate(frames):
        stamp = 100.0 + 0.1 * k
        if k == 2:
            oo.backend_odometry(stamp - 0.05)
        eo = pyorc.extract_features(fo, pyorc.project(fo, fr.xyz, fr.ring))
        rco, pose_o, R_o, t_o = oo.cloud(stamp, eo["corner"], eo["surface"])
        for line, tag in ((lines[2 * k], "staged"), (lines[2 * k + 1], "fused")):
            tok = line.split()
            assert tok[0] == tag and int(tok[1]) == rco
            pose = np.array([float(v) for v in tok[2:8]]); t = np.array([float(v) for v in tok[8:11]])
            assert int(tok[11]) == eo["corner"].shape[0] and int(tok[12]) == eo["surface"].shape[0]
            assert np.abs(pose[:3] - pose_o[:3]).max() <= 1e-4 and np.abs(pose[3:] - pose_o[3:]).max() <= 1e-5
            assert np.abs(t - t_o).max() <= 1e-4


def test_fused_path_with_deskew_equals_staged():
    from rolo_amd.frontend import deskew_params
    cfg = dict(n_scan=16, horizon_scan=1800)
    fg = front_params(**cfg)
    staged = LidarOdometry(0, 0.3); fe = FrontEnd(staged.reg, fg)
    fused = LidarOdometry(0, 0.3)
    for k, (R, t) in enumerate(trajectory(4)):
        fr = synth.make_frame("vlp16", R, t, synth.SEED + k)
        n = fr.xyz.shape[0]
        rel_time = (np.arange(n, dtype=np.float64) / n * 0.1).astype(np.float32)
        dsk = deskew_params([0.003, -0.002, 0.03], 0.1, 0.1)
        stamp = 100.0 + 0.1 * k
        if k == 2:
            staged.odometryHandler(stamp - 0.05); fused.odometryHandler(stamp - 0.05)
        fe.setDeskew(dsk, rel_time)
        pg = fe.project(fr.xyz, fr.ring); eg = fe.extract(pg["n"])
        rcs, pose_s, R_s, t_s = staged.cloudHandler(stamp, eg["corner"], eg["surface"])
        fused.setDeskew(dsk, rel_time)
        rcf, pose_f, R_f, t_f, cnt = fused.frame(fg, stamp, fr.xyz, fr.ring)
        assert rcf == rcs and cnt == (pg["n"], eg["corner"].shape[0], eg["surface"].shape[0])
        assert np.abs(pose_f - pose_s).max() < 1e-6 and np.abs(R_f - R_s).max() < 1e-9 and np.abs(t_f - t_s).max() < 1e-9


def _pack_msg(fr, kind):
    """Payload of a sensor_msgs/PointCloud2 as the drivers lay it out: Velodyne packed (point_step 22: x y z intensity
    float32 @0..12, ring uint16 @16, time float32 @18) or Ouster (point_step 48: x y z @0..8, intensity @16, t uint32 ns @20,
    reflectivity uint16 @24, ring uint8 @26, noise uint16 @28, range uint32 @32)."""
    from rolo_amd._lib import CloudLayout
    n = fr.xyz.shape[0]
    t = (np.arange(n, dtype=np.float64) / n * 0.1)
    if kind == "velodyne":
        dt = np.dtype({"names": ["x", "y", "z", "intensity", "ring", "time"], "formats": ["<f4", "<f4", "<f4", "<f4", "<u2", "<f4"],
                       "offsets": [0, 4, 8, 12, 16, 18], "itemsize": 22})
        a = np.zeros(n, dt); a["time"] = t.astype(np.float32)
        L = CloudLayout(22, 0, 4, 8, 16, 2, 18, 1)
        rel = np.abs(a["time"])
    else:
        dt = np.dtype({"names": ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "noise", "range"],
                       "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<u2", "u1", "<u2", "<u4"], "offsets": [0, 4, 8, 16, 20, 24, 26, 28, 32], "itemsize": 48})
        a = np.zeros(n, dt); a["t"] = (t * 1e9).astype(np.uint32)
        L = CloudLayout(48, 0, 4, 8, 26, 1, 20, 2)
        rel = np.abs(a["t"].astype(np.float32) * np.float32(1e-9))          # dst.time = src.t * 1e-9f, imageProjection.cpp:209
    a["x"], a["y"], a["z"] = fr.xyz[:, 0], fr.xyz[:, 1], fr.xyz[:, 2]
    a["intensity"] = 7.0; a["ring"] = fr.ring
    return a.view(np.uint8).reshape(-1), L, rel.astype(np.float32)


@pytest.mark.parametrize("kind", ["velodyne", "ouster"])
def test_submit_from_pointcloud2_payload(kind):
    """rolo_odom_submit_msg: the field extraction of cachePointCloud (pcl::moveFromROSMsg, the Ouster conversion loop) as a
    kernel on the raw message bytes — same frames, same poses as handing over x y z / ring arrays, with and without de-skew
    from the message's own times."""
    from rolo_amd.frontend import deskew_params
    cfg = dict(n_scan=16, horizon_scan=1800)
    fg = front_params(**cfg)
    arrays = LidarOdometry(0, 0.3); msgs = LidarOdometry(0, 0.3); dev = LidarOdometry(0, 0.3)
    for k, (R, t) in enumerate(trajectory(4)):
        fr = synth.make_frame("vlp16", R, t, synth.SEED + k)
        payload, L, rel = _pack_msg(fr, kind)
        stamp = 100.0 + 0.1 * k
        if k == 2:
            arrays.odometryHandler(stamp - 0.05); msgs.odometryHandler(stamp - 0.05)
        if k >= 2:   # de-skew the registered frames
            dsk = deskew_params([0.002, -0.001, 0.025], 0.1, 0.1)
            arrays.setDeskew(dsk, rel); msgs.setDeskewFromMessage(dsk)
        arrays.submit(fg, stamp, fr.xyz, fr.ring); msgs.submit_msg(fg, stamp, payload, L)
        ra, pa, Ra, ta, ca = arrays.collect(); rm, pm, Rm, tm, cm = msgs.collect()
        assert ra == rm and ca == cm
        assert np.abs(pm - pa).max() < 1e-6 and np.abs(Rm - Ra).max() < 1e-9 and np.abs(tm - ta).max() < 1e-9
        # the payload may already be in HBM (e.g. written by a driver with GPU-direct): same again from a device pointer
        import torch
        d_payload = torch.from_numpy(payload.copy()).cuda(); torch.cuda.synchronize()
        if k >= 2:
            dev.setDeskewFromMessage(dsk)
        if k == 2:
            dev.odometryHandler(stamp - 0.05)
        dev.submit_msg(fg, stamp, d_payload.data_ptr(), L, n_points=fr.xyz.shape[0])
        rd, pd, Rd, td, cd = dev.collect()
        assert rd == rm and cd == cm and np.abs(pd - pm).max() < 1e-6 and np.abs(td - tm).max() < 1e-9
    # a layout that points outside the record is refused
    from rolo_amd._lib import CloudLayout, RoloError
    with pytest.raises(RoloError) as ei:
        msgs.submit_msg(fg, 101.0, payload, CloudLayout(L.point_step, 0, 4, L.point_step - 2, L.off_ring, L.ring_bytes, L.off_time, L.time_kind))
    assert ei.value.code == -1


def long_trajectory(n):
    """a closed curve through the hall with changing speed and turn rate (0.15 ... 0.45 m and 1.5 ... 4 degrees per frame, a little roll / pitch): the view, and with it
    the number of corner / surface features, changes every frame"""
    poses = []
    R = np.eye(3); t = np.array([-4.0, -3.0, 0.0])
    for k in range(n):
        poses.append((R.copy(), t.copy()))
        step = 0.30 + 0.15 * np.sin(0.37 * k)
        yaw = np.deg2rad(2.75 + 1.25 * np.cos(0.23 * k))
        dR = synth.rpy_to_R(np.deg2rad(0.4 * np.sin(0.5 * k)), np.deg2rad(-0.3 * np.cos(0.31 * k)), yaw)
        t = t + R @ np.array([step, 0.03 * np.sin(0.9 * k), 0.0])
        R = R @ dR
    return poses


def test_pipeline_50_frames():
    from rolo_amd._lib import lib
    """A LONG sequence (round 5's verdict, item 5): 50 raw VLP-16 frames along a curved trajectory, feature counts changing every frame, through the staged node cores and
    the fused submit / collect path against the oracle chain (src/lidarOdometry.cpp:503-626): every state identical, every pose within the bars, and what only a long run
    shows — device buffers stop growing, the frame's hipGraph is re-captured a handful of times (the feature counts are part of its key), no top-up storms, and the
    schedule hints settle."""
    cfg = dict(n_scan=16, horizon_scan=1800)
    n = 50
    poses = long_trajectory(n)
    fo = pyorc.front_params(**cfg); fg = front_params(**cfg)
    oo = pyorc.Odom(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0)), 0.3)
    staged = LidarOdometry(0, 0.3); fe = FrontEnd(staged.reg, fg)
    fused = LidarOdometry(0, 0.3)
    frames = [synth.make_frame("vlp16", R, t, synth.SEED + 7 * k) for k, (R, t) in enumerate(poses)]
    fused.submit(fg, 100.0, frames[0].xyz, frames[0].ring)
    counts, counters = [], []
    worst = np.zeros(4)
    for k in range(n):
        fr = frames[k]; stamp = 100.0 + 0.1 * k
        if k >= 2:   # the back end answers every frame from the third on
            oo.backend_odometry(stamp - 0.05); staged.odometryHandler(stamp - 0.05); fused.odometryHandler(stamp - 0.05)
        eo = pyorc.extract_features(fo, pyorc.project(fo, fr.xyz, fr.ring))
        rco, pose_o, R_o, t_o = oo.cloud(stamp, eo["corner"], eo["surface"])
        pg = fe.project(fr.xyz, fr.ring); eg = fe.extract(pg["n"])
        assert np.array_equal(eg["corner"], eo["corner"]) and np.array_equal(eg["surface"], eo["surface"])
        rcs, pose_s, R_s, t_s = staged.cloudHandler(stamp, eg["corner"], eg["surface"])
        if k + 1 < n:
            fused.submit(fg, stamp + 0.1, frames[k + 1].xyz, frames[k + 1].ring)
        rcf, pose_f, R_f, t_f, cnt = fused.collect()
        assert rcs == rcf == rco, (k, rcs, rcf, rco)
        assert cnt == (pg["n"], eg["corner"].shape[0], eg["surface"].shape[0])
        for (p_, R_, t_) in ((pose_s, R_s, t_s), (pose_f, R_f, t_f)):
            assert np.abs(p_[:3] - pose_o[:3]).max() <= 1e-4 and np.abs(p_[3:] - pose_o[3:]).max() <= 1e-5, k
            assert np.abs(R_ - R_o).max() <= 1e-5 and np.abs(t_ - t_o).max() <= 1e-4, k
            worst = np.maximum(worst, [np.abs(p_[:3] - pose_o[:3]).max(), np.abs(p_[3:] - pose_o[3:]).max(), np.abs(R_ - R_o).max(), np.abs(t_ - t_o).max()])
        counts.append(cnt[1] + cnt[2])
        counters.append((staged.reg.counters(), fused.reg.counters(), int(lib().rolo_alloc_count())))
    assert len(set(counts)) > 25          # the feature counts did change from frame to frame
    assert worst.max() < 5e-6, worst      # what we actually get over 50 frames (float32 pose chain): no drift between the chains
    for which in (0, 1):
        c = [cc[which] for cc in counters]
        assert c[-1]["frames"] == n - 2                       # the first frame is only stored, the second waits for the back end's first odometry
        # a frame's hipGraph is keyed on its cloud sizes: with feature counts changing every frame almost nothing is replayed — but nothing may storm either
        assert c[-1]["topup_frames"] <= 3, c[-1]
        assert c[-1]["sync_chunks"] <= 8, c[-1]
        assert c[-1]["hint_rot"] == c[20]["hint_rot"] and c[-1]["hint_trans"] <= c[20]["hint_trans"] + 2   # the schedule hints settled in the first twenty frames
    # device allocations stop: buffers grow by a quarter beyond what a frame needs, so the count of (re)allocations is the same after frame 25 and after frame 50
    assert counters[-1][2] == counters[25][2], (counters[25][2], counters[-1][2])
    staged.close(); fused.close()
