"""Host mirror of the back end's scan-to-submap optimisation (reference src/backMapping.cpp:681-1058) over rolo_scan2map_optimize."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import lib, check, Scan2MapStats
from .rotvgicp import RotVGICP


class Scan2Map:
    def __init__(self, device: int = 0, edgeFeatureMinValidNum: int = 10, surfFeatureMinValidNum: int = 100):
        self.reg = RotVGICP(device)   # a context of its own: its clouds hold the sub-map trees
        self.edge_min, self.surf_min = edgeFeatureMinValidNum, surfFeatureMinValidNum
        self.last_stats = None

    def close(self):
        self.reg.close()

    def setSubmap(self, map_corner, map_surf):
        """kdtree*FromMap->setInputCloud (:690-691): the sub-map stays resident; scan2MapOptimization(corner, surf, None, None, tf) registers against it"""
        fp = C.POINTER(C.c_float)
        a = [np.ascontiguousarray(x, np.float32).reshape(-1, 4) for x in (map_corner, map_surf)]
        check(lib().rolo_scan2map_set_submap(self.reg._h, a[0].ctypes.data_as(fp), a[0].shape[0], a[1].ctypes.data_as(fp), a[1].shape[0]), "rolo_scan2map_set_submap")

    def scan2MapOptimization(self, corner, surf, map_corner, map_surf, transformTobeMapped, want_debug=False):
        """Returns the updated transformTobeMapped (roll, pitch, yaw, x, y, z; float32) [, selected flags, coeffSel of the last iteration].
        map_corner = map_surf = None: the resident sub-map of setSubmap."""
        fp = C.POINTER(C.c_float)
        if map_corner is None and map_surf is None:
            map_corner = map_surf = np.zeros((0, 4), np.float32)
            resident = True
        else:
            resident = False
        a = [np.ascontiguousarray(x, np.float32).reshape(-1, 4) for x in (corner, surf, map_corner, map_surf)]
        tf = np.ascontiguousarray(transformTobeMapped, np.float32).copy()
        st = Scan2MapStats()
        n = a[0].shape[0] + a[1].shape[0]
        sel = np.zeros(max(n, 1), np.uint8) if want_debug else None
        coeff = np.zeros((max(n, 1), 4), np.float32) if want_debug else None
        check(lib().rolo_scan2map_optimize(self.reg._h, a[0].ctypes.data_as(fp), a[0].shape[0], a[1].ctypes.data_as(fp), a[1].shape[0], None if resident else a[2].ctypes.data_as(fp), a[2].shape[0],
                                           None if resident else a[3].ctypes.data_as(fp), a[3].shape[0], tf.ctypes.data_as(fp), self.edge_min, self.surf_min, C.byref(st),
                                           sel.ctypes.data_as(C.POINTER(C.c_ubyte)) if want_debug else None, coeff.ctypes.data_as(fp) if want_debug else None),
              "rolo_scan2map_optimize")
        self.last_stats = st
        return (tf, sel[:n].astype(bool), coeff[:n]) if want_debug else tf
