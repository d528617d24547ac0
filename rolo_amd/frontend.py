"""Host-side mirror of the reference front-end node cores over the C ABI:

  ImageProjection.projectPointCloud + cloudExtraction   (reference src/imageProjection.cpp:399-505)
  FeatureExtraction.calculateSmoothness / markOccludedPoints / extractFeatures (src/featureExtraction.cpp:87-266)

The arrays mirror the fields of rolo/CloudInfoStamp (msg/CloudInfoStamp.msg): startRingIndex, endRingIndex,
pointColInd, pointRange, cloud_projected, extracted_corner, extracted_surface. Plumbing only.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from ._lib import Deskew, FrontParams, check, lib


def front_params(**kw) -> FrontParams:
    p = FrontParams()
    lib().rolo_front_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def deskew_params(odom_incre_rpy, scan_period=0.1, odom_time_diff=0.1, enabled=True) -> Deskew:
    """odomIncreRoll/Pitch/Yaw, scanPeriod, odomTimeDiff of ImageProjection (imageProjection.cpp:79-81,349-351)."""
    d = Deskew(); d.enabled = int(enabled); d.scan_period = scan_period; d.odom_time_diff = odom_time_diff
    for k in range(3):
        d.odom_incre_rpy[k] = odom_incre_rpy[k]
    return d


def odom_increment(front6, back6):
    """lidarOdomAffineFront.inverse() * lidarOdomAffineBack as x, y, z, roll, pitch, yaw (imageProjection.cpp:345-351)."""
    f = np.ascontiguousarray(front6, np.float32); b = np.ascontiguousarray(back6, np.float32); o = np.zeros(6, np.float32)
    lib().rolo_odom_increment(_f(f), _f(b), _f(o))
    return o


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class FrontEnd:
    """Owns nothing but a reference to a RotVGICP context (its device buffers and stream are reused)."""

    def __init__(self, ctx, params: FrontParams):
        self.ctx = ctx
        self.p = params

    def setDeskew(self, dsk: Deskew, rel_time):
        """deskewPoint for the next project(): rel_time[i] = fabs(point.time) of raw point i (numpy array)."""
        rt = np.ascontiguousarray(rel_time, np.float32)
        check(lib().rolo_front_set_deskew(self.ctx._h, C.byref(dsk), C.c_void_p(rt.ctypes.data), rt.shape[0], 0), "rolo_front_set_deskew")

    def setDeskewFromCloud(self, dsk: Deskew):
        """deskewPoint for the next project() with the azimuth-interpolated times of deskewCloudInfo's timeFlag == -1 branch."""
        check(lib().rolo_front_set_deskew(self.ctx._h, C.byref(dsk), None, 0, 0), "rolo_front_set_deskew")

    def project(self, xyz, ring, want_range_mat=False):
        xyz = np.ascontiguousarray(xyz, np.float32)
        ring = np.ascontiguousarray(ring, np.uint16)
        NS, H = self.p.n_scan, self.p.horizon_scan
        ext = np.zeros((NS * H, 4), np.float32); col = np.zeros(NS * H, np.int32); rng = np.zeros(NS * H, np.float32)
        sr = np.zeros(NS, np.int32); er = np.zeros(NS, np.int32)
        rm = np.zeros(NS * H, np.float32) if want_range_mat else None
        n = C.c_int(0)
        check(lib().rolo_project_frame(self.ctx._h, C.byref(self.p), _f(xyz), xyz.shape[1], ring.ctypes.data_as(C.POINTER(C.c_uint16)),
                                       xyz.shape[0], _f(ext), _i(col), _f(rng), _i(sr), _i(er), _f(rm) if rm is not None else None,
                                       C.byref(n)), "rolo_project_frame")
        n = n.value
        out = dict(n=n, extracted=ext[:n].copy(), point_col_ind=col[:n].copy(), point_range=rng[:n].copy(), start_ring=sr, end_ring=er)
        if rm is not None:
            out["range_mat"] = rm.reshape(NS, H)
        return out

    def extract(self, n, debug=False):
        corner = np.zeros((max(n, 1), 4), np.float32); surf = np.zeros((max(n, 1), 4), np.float32)
        nc = C.c_int(0); ns = C.c_int(0)
        curv = np.zeros(max(n, 1), np.float32) if debug else None
        picked = np.zeros(max(n, 1), np.int32) if debug else None
        label = np.zeros(max(n, 1), np.int32) if debug else None
        check(lib().rolo_extract_features(self.ctx._h, C.byref(self.p), _f(corner), C.byref(nc), _f(surf), C.byref(ns),
                                          _f(curv) if debug else None, _i(picked) if debug else None, _i(label) if debug else None),
              "rolo_extract_features")
        out = dict(corner=corner[:nc.value].copy(), surface=surf[:ns.value].copy())
        if debug:
            out.update(curvature=curv[:n], picked=picked[:n], label=label[:n])
        return out
