"""Independent Python statement of the ROS1 wire format (roscpp_serialization rules) for the messages of the front-end topic surface —
the cross-check of include/rolo_ros_wire.hpp. Little-endian; string = uint32 length + bytes; T[] = uint32 count + elements;
T[N] = elements only; bool = uint8; time = uint32 sec + uint32 nsec. Messages are plain dicts."""
import struct

import numpy as np

FLOAT32, UINT16, UINT8, UINT32 = 7, 4, 2, 6


def _str(s):
    b = s.encode()
    return struct.pack("<I", len(b)) + b


def pack_header(h):
    return struct.pack("<III", h.get("seq", 0), h["sec"], h["nsec"]) + _str(h.get("frame_id", ""))


def pack_pc2(m):
    out = pack_header(m["header"]) + struct.pack("<II", m["height"], m["width"]) + struct.pack("<I", len(m["fields"]))
    for name, off, dt, cnt in m["fields"]:
        out += _str(name) + struct.pack("<IBI", off, dt, cnt)
    data = bytes(m["data"])
    out += struct.pack("<BII", m.get("is_bigendian", 0), m["point_step"], m["row_step"]) + struct.pack("<I", len(data)) + data
    out += struct.pack("<B", m["is_dense"])
    return out


class _R:
    def __init__(self, b):
        self.b = memoryview(b); self.i = 0

    def u(self, fmt):
        n = struct.calcsize(fmt); v = struct.unpack_from(fmt, self.b, self.i); self.i += n
        return v if len(v) > 1 else v[0]

    def s(self):
        n = self.u("<I"); v = bytes(self.b[self.i:self.i + n]).decode(); self.i += n
        return v

    def arr(self, dtype):
        n = self.u("<I"); a = np.frombuffer(self.b, dtype, n, self.i).copy(); self.i += n * a.itemsize
        return a


def _header(r):
    seq, sec, nsec = r.u("<III")
    return dict(seq=seq, sec=sec, nsec=nsec, frame_id=r.s())


def _pc2(r):
    m = dict(header=_header(r))
    m["height"], m["width"] = r.u("<II")
    m["fields"] = []
    for _ in range(r.u("<I")):
        name = r.s(); off, dt, cnt = r.u("<IBI"); m["fields"].append((name, off, dt, cnt))
    m["is_bigendian"], m["point_step"], m["row_step"] = r.u("<BII")
    m["data"] = r.arr(np.uint8)
    m["is_dense"] = r.u("<B")
    return m


def parse_pc2(b):
    r = _R(b); m = _pc2(r); assert r.i == len(b)
    return m


def parse_cloud_info(b):
    r = _R(b)
    m = dict(header=_header(r))
    for k in ("startRingIndex", "endRingIndex", "pointColInd"):
        m[k] = r.arr(np.int32)
    m["pointRange"] = r.arr(np.float32)
    m["orientation"] = r.u("<fff")
    m["initialGuess"] = np.array(r.u("<ffffff"), np.float32)
    m["covariance"] = r.arr(np.float32)
    m["odomAvailable"] = r.u("<B")
    for k in ("cloud_projected", "extracted_corner", "extracted_surface", "extracted_normal", "extracted_ground"):
        m[k] = _pc2(r)
    assert r.i == len(b)
    return m


def parse_odometry(b):
    r = _R(b)
    m = dict(header=_header(r), child_frame_id=r.s())
    m["position"] = np.array(r.u("<ddd")); m["orientation"] = np.array(r.u("<dddd"))
    m["pose_covariance"] = np.array(r.u("<36d"))
    m["twist"] = np.array(r.u("<6d")); m["twist_covariance"] = np.array(r.u("<36d"))
    assert r.i == len(b)
    return m


def pack_odometry(m):
    """nav_msgs/Odometry: header, child_frame_id, PoseWithCovariance (7 + 36 doubles), TwistWithCovariance (6 + 36 doubles)"""
    return (pack_header(m["header"]) + _str(m.get("child_frame_id", "")) + struct.pack("<7d", *m["position"], *m["orientation"]) +
            struct.pack("<36d", *m.get("pose_covariance", [0.0] * 36)) + struct.pack("<6d", *m.get("twist", [0.0] * 6)) +
            struct.pack("<36d", *m.get("twist_covariance", [0.0] * 36)))


def parse_path(b):
    """nav_msgs/Path: header, PoseStamped[]"""
    r = _R(b)
    m = dict(header=_header(r), poses=[])
    for _ in range(r.u("<I")):
        h = _header(r)
        m["poses"].append(dict(header=h, position=np.array(r.u("<ddd")), orientation=np.array(r.u("<dddd"))))
    assert r.i == len(b)
    return m


def parse_pose_cov_stamped(b):
    """geometry_msgs/PoseWithCovarianceStamped"""
    r = _R(b)
    m = dict(header=_header(r), position=np.array(r.u("<ddd")), orientation=np.array(r.u("<dddd")), covariance=np.array(r.u("<36d")))
    assert r.i == len(b)
    return m


def xyzi_of(pc2):
    """pcl::fromROSMsg for PointXYZI records"""
    n = pc2["height"] * pc2["width"]
    rec = pc2["data"].reshape(n, pc2["point_step"]) if n else np.zeros((0, pc2["point_step"]), np.uint8)
    offs = {name: off for name, off, dt, cnt in pc2["fields"]}
    out = np.zeros((n, 4), np.float32)
    for c, name in enumerate(("x", "y", "z", "intensity")):
        out[:, c] = rec[:, offs[name]:offs[name] + 4].copy().view(np.float32)[:, 0] if n else 0
    return out


def velodyne_msg(frame, stamp, seq=0, frame_id="velodyne", with_time=True):
    """sensor_msgs/PointCloud2 as the Velodyne driver publishes it: x, y, z, intensity FLOAT32, ring UINT16 @16, time FLOAT32 @18; 22-byte records"""
    n = frame.xyz.shape[0]
    step = 22 if with_time else 18
    rec = np.zeros((n, step), np.uint8)
    rec[:, 0:12] = np.ascontiguousarray(frame.xyz, np.float32).view(np.uint8).reshape(n, 12)
    rec[:, 12:16] = np.ascontiguousarray(frame.intensity, np.float32).view(np.uint8).reshape(n, 4)
    rec[:, 16:18] = np.ascontiguousarray(frame.ring, np.uint16).view(np.uint8).reshape(n, 2)
    fields = [("x", 0, FLOAT32, 1), ("y", 4, FLOAT32, 1), ("z", 8, FLOAT32, 1), ("intensity", 12, FLOAT32, 1), ("ring", 16, UINT16, 1)]
    if with_time:
        rec[:, 18:22] = np.ascontiguousarray(frame.time, np.float32).view(np.uint8).reshape(n, 4)
        fields.append(("time", 18, FLOAT32, 1))
    sec = int(np.floor(stamp)); nsec = int(round((stamp - sec) * 1e9))
    return dict(header=dict(seq=seq, sec=sec, nsec=nsec, frame_id=frame_id), height=1, width=n, fields=fields, is_bigendian=0,
                point_step=step, row_step=step * n, data=rec.reshape(-1), is_dense=1)


def ouster_msg(frame, stamp, seq=0, frame_id="os_sensor"):
    """Ouster driver layout (imageProjection.cpp:21-35): x, y, z @0, intensity @16, t UINT32 @20 (ns), reflectivity UINT16 @24, ring UINT8 @26,
    noise UINT16 @28, range UINT32 @32; 48-byte records"""
    n = frame.xyz.shape[0]
    step = 48
    rec = np.zeros((n, step), np.uint8)
    rec[:, 0:12] = np.ascontiguousarray(frame.xyz, np.float32).view(np.uint8).reshape(n, 12)
    rec[:, 16:20] = np.ascontiguousarray(frame.intensity, np.float32).view(np.uint8).reshape(n, 4)
    rec[:, 20:24] = np.ascontiguousarray(np.round(frame.time.astype(np.float64) * 1e9), np.uint32).view(np.uint8).reshape(n, 4)
    rec[:, 26] = frame.ring.astype(np.uint8)
    fields = [("x", 0, FLOAT32, 1), ("y", 4, FLOAT32, 1), ("z", 8, FLOAT32, 1), ("intensity", 16, FLOAT32, 1), ("t", 20, UINT32, 1),
              ("reflectivity", 24, UINT16, 1), ("ring", 26, UINT8, 1), ("noise", 28, UINT16, 1), ("range", 32, UINT32, 1)]
    sec = int(np.floor(stamp)); nsec = int(round((stamp - sec) * 1e9))
    return dict(header=dict(seq=seq, sec=sec, nsec=nsec, frame_id=frame_id), height=1, width=n, fields=fields, is_bigendian=0,
                point_step=step, row_step=step * n, data=rec.reshape(-1), is_dense=1)
