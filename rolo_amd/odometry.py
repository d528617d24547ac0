"""Host-side mirror of the reference node core `LidarOdometry` (reference src/lidarOdometry.cpp:325-713) over the
C ABI (rolo_odom_*): same state machine — first frame stored, gated until the back end has published once
(odometryHandler), then forward prediction + RotVGICP rotation + continuous-time translation per frame.
Plumbing only; the pose algebra and the registration live in librolo_hip.so."""
from __future__ import annotations

import ctypes as C
import numpy as np

from ._lib import check, lib
from .rotvgicp import RotVGICP


class LidarOdometry:
    def __init__(self, device: int = 0, ct_lambda: float = 0.3, polar_resolution=(0.175, 0.175, 2.0)):
        self.reg = RotVGICP(device)
        self.reg.setPolarResolution(*polar_resolution)  # lidarOdometry.cpp:462
        # (one launch per LM trial is the driver's own option, ROLO_ODOM_FUSED_LM, on by default: nothing to set on the operator)
        h = C.c_void_p()
        check(lib().rolo_odom_create(self.reg._h, ct_lambda, C.byref(h)), "rolo_odom_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().rolo_odom_destroy(self._h)
            self._h = None
        self.reg.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    REUSE_COVARIANCES = 1
    FUSED_LM = 2
    EARLY_SOURCE = 3

    def setDeskew(self, dsk, rel_time):
        """deskewPoint for the next submit() / frame(): rel_time[i] = fabs(point.time) of raw point i (numpy array)."""
        rt = np.ascontiguousarray(rel_time, np.float32)
        check(lib().rolo_odom_set_deskew(self._h, C.byref(dsk), C.c_void_p(rt.ctypes.data), rt.shape[0], 0), "rolo_odom_set_deskew")

    def setOption(self, option: int, value: int):
        check(lib().rolo_odom_set_option(self._h, option, value), "rolo_odom_set_option")

    def frame(self, front_params, stamp: float, xyz, ring, n_raw=None, stride=None):
        """Fused device-resident path (rolo_odom_frame): raw frame -> pose. `xyz` / `ring` are numpy arrays (host) or
        integer device pointers (then pass n_raw and stride). Returns (status, pose6, Rotation, Translation, (N, n_corner, n_surface))."""
        pose = np.zeros(6, np.float32); R = np.zeros((3, 3)); t = np.zeros(3); counts = (C.c_int * 3)()
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        if isinstance(xyz, int):
            pp, rp, on_dev = C.c_void_p(xyz), C.c_void_p(ring), 1
        else:
            xyz = np.ascontiguousarray(xyz, np.float32); ring = np.ascontiguousarray(ring, np.uint16)
            n_raw, stride = xyz.shape[0], xyz.shape[1]
            pp, rp, on_dev = C.c_void_p(xyz.ctypes.data), C.c_void_p(ring.ctypes.data), 0
        rc = check(lib().rolo_odom_frame(self._h, C.byref(front_params), stamp, pp, stride, rp, n_raw, on_dev, pose.ctypes.data_as(fp),
                                         R.ctypes.data_as(dp), t.ctypes.data_as(dp), counts), "rolo_odom_frame")
        return rc, pose, R, t, tuple(counts)

    def submit(self, front_params, stamp: float, xyz, ring, n_raw=None, stride=None):
        """First half of frame(): enqueue projection + feature extraction on the driver's front-end stream (returns at once;
        up to two frames in flight). Host arrays are kept alive until the matching collect()."""
        if isinstance(xyz, int):
            pp, rp, on_dev = C.c_void_p(xyz), C.c_void_p(ring), 1
            keep = None   # placeholder: _keep stays 1:1 with the frames in flight
        else:
            xyz = np.ascontiguousarray(xyz, np.float32); ring = np.ascontiguousarray(ring, np.uint16)
            n_raw, stride = xyz.shape[0], xyz.shape[1]
            pp, rp, on_dev = C.c_void_p(xyz.ctypes.data), C.c_void_p(ring.ctypes.data), 0
            keep = (xyz, ring)
        self._keep = getattr(self, "_keep", []) + [keep]
        check(lib().rolo_odom_submit(self._h, C.byref(front_params), stamp, pp, stride, rp, n_raw, on_dev), "rolo_odom_submit")

    def submit_msg(self, front_params, stamp: float, payload, layout, n_points=None):
        """submit() from the bytes of a sensor_msgs/PointCloud2: `payload` a uint8 array (n * point_step) or an integer device
        pointer (then pass n_points), `layout` a _lib.CloudLayout with the field offsets of the message."""
        if isinstance(payload, int):
            ptr, n, on_dev = C.c_void_p(payload), n_points, 1
            keep = None
        else:
            payload = np.ascontiguousarray(payload, np.uint8)
            ptr, n, on_dev = C.c_void_p(payload.ctypes.data), payload.size // layout.point_step, 0
            keep = (payload,)
        self._keep = getattr(self, "_keep", []) + [keep]
        check(lib().rolo_odom_submit_msg(self._h, C.byref(front_params), stamp, ptr, C.byref(layout), n, on_dev), "rolo_odom_submit_msg")

    def setDeskewFromMessage(self, dsk):
        """De-skew the next submit_msg() with the per-point times of the message itself."""
        check(lib().rolo_odom_set_deskew(self._h, C.byref(dsk), None, 0, 0), "rolo_odom_set_deskew")

    def features(self):
        """(corner, surface) clouds of the last collected frame of the fused path (n x 4: x, y, z, intensity)."""
        nc, ns = C.c_int(0), C.c_int(0)
        check(lib().rolo_odom_get_features(self._h, None, 0, C.byref(nc), C.byref(ns)), "rolo_odom_get_features")
        buf = np.zeros((max(nc.value + ns.value, 1), 4), np.float32)
        check(lib().rolo_odom_get_features(self._h, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.shape[0], C.byref(nc), C.byref(ns)), "rolo_odom_get_features")
        return buf[:nc.value].copy(), buf[nc.value:nc.value + ns.value].copy()

    def collect(self):
        """Second half: finish the oldest submitted frame. Returns what frame() returns."""
        pose = np.zeros(6, np.float32); R = np.zeros((3, 3)); t = np.zeros(3); counts = (C.c_int * 3)()
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        rc = check(lib().rolo_odom_collect(self._h, pose.ctypes.data_as(fp), R.ctypes.data_as(dp), t.ctypes.data_as(dp), counts), "rolo_odom_collect")
        if getattr(self, "_keep", None):   # host inputs of the collected frame may go (one entry per frame in flight)
            self._keep.pop(0)
        return rc, pose, R, t, tuple(counts)

    def odometryHandler(self, stamp: float):
        check(lib().rolo_odom_backend_odometry(self._h, stamp), "rolo_odom_backend_odometry")

    def cloudHandler(self, stamp: float, corner: np.ndarray, surface: np.ndarray):
        """Returns (status, LaserOdomPose[6] float32, Rotation 3x3, Translation 3)."""
        corner = np.ascontiguousarray(corner, np.float32).reshape(-1, 4)
        surface = np.ascontiguousarray(surface, np.float32).reshape(-1, 4)
        pose = np.zeros(6, np.float32); R = np.zeros((3, 3)); t = np.zeros(3)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        rc = check(lib().rolo_odom_cloud(self._h, stamp, corner.ctypes.data_as(fp), corner.shape[0], surface.ctypes.data_as(fp),
                                         surface.shape[0], pose.ctypes.data_as(fp), R.ctypes.data_as(dp), t.ctypes.data_as(dp)),
                   "rolo_odom_cloud")
        return rc, pose, R, t
