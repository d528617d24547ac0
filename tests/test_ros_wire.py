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
