"""CPU: the ROS1 wire (de)serialisers of include/rolo_ros_wire.hpp against an independent Python statement of the format
(tests/ros1_wire.py) and hand-written bytes; tf::createQuaternionFromRPY / getRPY against scipy. No GPU: the C++ demo only touches
the wire layer in these modes."""
import os
import struct
import subprocess

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from rolo_amd import synth
from tests import ros1_wire as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("wire") / "ros_wire_demo")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "ros_wire_demo.cpp"), "-o", exe,
           "-L", os.path.join(ROOT, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(ROOT, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def test_known_answer_bytes_of_a_small_pointcloud2(demo, tmp_path):
    # hand-assembled message: header(seq 7, stamp 12.5 s, "ab"), 1 x 2 points of 8 bytes (x FLOAT32 @0, ring UINT16 @4), dense
    raw = (struct.pack("<III", 7, 12, 500000000) + struct.pack("<I", 2) + b"ab" + struct.pack("<II", 1, 2) + struct.pack("<I", 2) +
           struct.pack("<I", 1) + b"x" + struct.pack("<IBI", 0, 7, 1) + struct.pack("<I", 4) + b"ring" + struct.pack("<IBI", 4, 4, 1) +
           struct.pack("<BII", 0, 8, 16) + struct.pack("<I", 16) + struct.pack("<fHHfHH", 1.5, 3, 0, -2.0, 9, 0) + struct.pack("<B", 1))
    m = W.parse_pc2(raw)
    assert m["header"] == dict(seq=7, sec=12, nsec=500000000, frame_id="ab") and m["fields"] == [("x", 0, 7, 1), ("ring", 4, 4, 1)]
    assert W.pack_pc2(m) == raw
    (tmp_path / "in.bin").write_bytes(raw)
    r = subprocess.run([demo, "roundtrip", "pc2", str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split()[0] == "ok"
    assert (tmp_path / "out.bin").read_bytes() == raw
    # a truncated message is refused, and so is one with trailing bytes
    for bad in (raw[:-3], raw + b"\0"):
        (tmp_path / "bad.bin").write_bytes(bad)
        r = subprocess.run([demo, "roundtrip", "pc2", str(tmp_path / "bad.bin"), str(tmp_path / "o.bin")], capture_output=True, text=True)
        assert r.returncode == 3 and r.stdout.strip() == "malformed"


def test_velodyne_and_ouster_messages_round_trip(demo, tmp_path):
    fr = synth.make_frame("vlp16", np.eye(3), np.zeros(3), synth.SEED, col_stride=16)
    for k, msg in enumerate((W.velodyne_msg(fr, 100.25, seq=3), W.ouster_msg(fr, 7.0))):
        raw = W.pack_pc2(msg)
        (tmp_path / f"m{k}.bin").write_bytes(raw)
        r = subprocess.run([demo, "roundtrip", "pc2", str(tmp_path / f"m{k}.bin"), str(tmp_path / f"o{k}.bin")], capture_output=True, text=True)
        assert r.returncode == 0
        assert (tmp_path / f"o{k}.bin").read_bytes() == raw


def test_cloud_info_and_odometry_layouts(demo, tmp_path):
    # CloudInfoStamp assembled by hand in msg/CloudInfoStamp.msg order
    def pc2(n):
        return W.pack_pc2(dict(header=dict(seq=0, sec=1, nsec=2, frame_id="l"), height=1, width=n, fields=[("x", 0, 7, 1)], is_bigendian=0,
                               point_step=4, row_step=4 * n, data=np.arange(4 * n, dtype=np.uint8), is_dense=1))
    hdr = struct.pack("<III", 1, 2, 3) + struct.pack("<I", 1) + b"f"
    arr_i = lambda v: struct.pack("<I", len(v)) + np.asarray(v, np.int32).tobytes()
    arr_f = lambda v: struct.pack("<I", len(v)) + np.asarray(v, np.float32).tobytes()
    raw = (hdr + arr_i([4, 5]) + arr_i([6, 7]) + arr_i([1, 2, 3]) + arr_f([0.5, 1.5, 2.5]) + struct.pack("<fff", 0.1, 0.2, 0.3) +
           struct.pack("<ffffff", 1, 2, 3, 4, 5, 6) + arr_f([9.0]) + struct.pack("<B", 1) + pc2(3) + pc2(1) + pc2(2) + pc2(0) + pc2(0))
    m = W.parse_cloud_info(raw)
    assert list(m["pointColInd"]) == [1, 2, 3] and m["odomAvailable"] == 1 and list(m["initialGuess"]) == [1, 2, 3, 4, 5, 6]
    assert m["extracted_surface"]["width"] == 2
    (tmp_path / "c.bin").write_bytes(raw)
    r = subprocess.run([demo, "roundtrip", "cis", str(tmp_path / "c.bin"), str(tmp_path / "co.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "co.bin").read_bytes() == raw
    # nav_msgs/Odometry: header, child_frame_id, 7 + 36 + 6 + 36 doubles
    odo = (hdr + struct.pack("<I", 3) + b"lid" + struct.pack("<7d", 1, 2, 3, 0, 0, 0, 1) + struct.pack("<36d", *range(36)) +
           struct.pack("<6d", *range(6)) + struct.pack("<36d", *range(36)))
    assert W.parse_odometry(odo)["child_frame_id"] == "lid"
    (tmp_path / "o.bin").write_bytes(odo)
    r = subprocess.run([demo, "roundtrip", "odom", str(tmp_path / "o.bin"), str(tmp_path / "oo.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "oo.bin").read_bytes() == odo


@pytest.mark.parametrize("rpy", [(0.1, -0.2, 1.3), (0.0, 0.0, 0.0), (-3.0, 1.2, 2.9), (1e-4, -2e-4, 3e-4)])
def test_quaternion_from_rpy_matches_scipy(demo, rpy):
    r = subprocess.run([demo, "quat", *[repr(v) for v in rpy]], capture_output=True, text=True, check=True)
    v = np.array([float(x) for x in r.stdout.split()])
    q, back = v[:4], v[4:]
    want = Rotation.from_euler("xyz", rpy).as_quat()   # extrinsic x, y, z = Rz(yaw) Ry(pitch) Rx(roll); (x, y, z, w)
    assert min(np.abs(q - want).max(), np.abs(q + want).max()) < 1e-15
    assert np.abs(Rotation.from_euler("xyz", back).as_matrix() - Rotation.from_euler("xyz", rpy).as_matrix()).max() < 1e-14


def test_transform_fusion_node_bytes_on_the_reference_bag(demo, tmp_path):
    """The TransformFusion half of the rolo_lidarOdometry node (src/lidarOdometry.cpp:47-323) at message level, no GPU: the head of the
    reference's own resource/test_odom.bag as serialized nav_msgs/Odometry on odomTopic + "_incremental", the back end's message twice, timer
    ticks in between — and the serialized odomTopic / speed / path / future_pose_lidar messages against the independent twin's fixture
    (tests/golden/test_odom_bag_fusion.npz, made by tests/golden/make_golden_eskf.py)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "test_odom_bag.npz")); f = np.load(os.path.join(ROOT, "tests", "golden", "test_odom_bag_fusion.npz"))
    n = int(f["n"])
    blob = b""
    for k in range(n):
        m = W.pack_odometry(dict(header=dict(seq=k, sec=int(z["sec"][k]), nsec=int(z["nsec"][k]), frame_id="odom"), child_frame_id="lidar_odometry",
                                 position=z["pose"][k, :3], orientation=z["pose"][k, 3:], pose_covariance=np.arange(36.0), twist=[0, 0, 0, 0.1, 0.2, 0.3]))
        blob += struct.pack("<I", len(m)) + m
    (tmp_path / "msgs.bin").write_bytes(blob)
    (tmp_path / "backend.bin").write_bytes(W.pack_odometry(dict(header=dict(seq=0, sec=0, nsec=0, frame_id="map"), position=f["backend_p"], orientation=f["backend_q"])))
    out = tmp_path / "out"; out.mkdir()
    r = subprocess.run([demo, "fusion", str(tmp_path / "backend.bin"), str(out), ",".join(str(int(a)) for a in f["mapping_at"]), str(tmp_path / "msgs.bin")],
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split()[:3] == ["ok", str(n), "messages"], r.stdout + r.stderr
    want = f["ticks"]
    scale = max(1.0, float(np.abs(z["pose"][:n, :3]).max()))
    raw = (out / "fused.bin").read_bytes(); i = 0; got = []
    while i < len(raw):
        tick, ln = struct.unpack_from("<II", raw, i); i += 8
        got.append((tick, W.parse_odometry(raw[i:i + ln]))); i += ln
    assert [t for t, _ in got] == [int(t) for t in want[:, 0]]            # the same ticks publish / return early
    sp = np.frombuffer((out / "speed.bin").read_bytes(), dtype=[("tick", "<u4"), ("v", "<f4")])
    pl = np.frombuffer((out / "path_len.bin").read_bytes(), dtype="<u4").reshape(-1, 2)
    assert len(sp) == len(got) == len(pl)
    for j, (tick, m) in enumerate(got):
        w = want[j]
        now = m["header"]["sec"] + 1e-9 * m["header"]["nsec"]
        assert abs(now - w[1]) < 1e-9 and m["header"]["frame_id"] == "odom" and m["child_frame_id"] == "base_link"
        assert m["header"]["seq"] == tick // 2                                # the template: the latest odomTopic + "_incremental" message ...
        assert np.array_equal(m["pose_covariance"], np.arange(36.0)) and np.array_equal(m["twist"][3:], [0.1, 0.2, 0.3])   # ... incl. what the handler does not touch
        # float Affine3f chain (front^-1 * back, then mapping * increment) on coordinates of up to ~70 m: a few float32 ulps of THOSE
        assert np.abs(m["position"] - w[2:5]).max() < 6e-7 * scale
        Ra = Rotation.from_quat(m["orientation"]).as_matrix(); Rb = Rotation.from_quat(w[5:9]).as_matrix()
        assert np.abs(Ra - Rb).max() < 2e-6
        assert np.abs(m["twist"][:3] - w[9:12]).max() < 1e-5 and abs(float(sp["v"][j]) - w[12]) < 1e-5
        assert bool(pl[j, 1] >> 31) == bool(w[13]) and int(pl[j, 1] & 0x7fffffff) == int(w[14])
    # future_pose_lidar: the last propagated pose in the lidar frame
    raw = (out / "future.bin").read_bytes(); i = 0; fut = []
    while i < len(raw):
        k, npts, ln = struct.unpack_from("<III", raw, i); i += 12
        fut.append((k, npts, W.parse_pose_cov_stamped(raw[i:i + ln]))); i += ln
    wf = f["future"]
    assert [(k, npts) for k, npts, _ in fut] == [(int(a[0]), int(a[1])) for a in wf]
    for (k, npts, m), a in zip(fut, wf):
        assert m["header"]["frame_id"] == "lidar_link" and not m["covariance"].any()
        assert np.abs(m["position"] - a[2:5]).max() < 1e-5 * max(1.0, 0.2 * npts) and m["position"][2] == 0.0
        # (the future pose lies 8 m of constant-jerk propagation ahead: a 1e-7 rad difference of a measured rotation — Eigen's float polar factor
        # here, a double orthonormalisation in the twin — becomes about 1e-6 rad/s^2
        # of angular acceleration: the tolerance grows with the SQUARE of the propagated time, 0.2 s per point (constant-jerk model; at crawling
        # speed the 8 m horizon is half a minute away). The strict check of the filter is the odomTopic message above.)
        assert np.abs(Rotation.from_quat(m["orientation"]).as_matrix() - Rotation.from_quat(a[5:9]).as_matrix()).max() < 2e-6 * max(1.0, (0.2 * npts) ** 2)


def test_path_and_small_messages_round_trip():
    """nav_msgs/Path, std_msgs/Float32 layouts by hand"""
    hdr = struct.pack("<III", 1, 2, 3) + struct.pack("<I", 1) + b"o"
    pose = struct.pack("<7d", 1, 2, 3, 0, 0, 0, 1)
    raw = hdr + struct.pack("<I", 2) + hdr + pose + hdr + pose
    m = W.parse_path(raw)
    assert len(m["poses"]) == 2 and m["poses"][1]["position"][2] == 3.0
