"""ctypes binding of oracle/librolo_oracle.so — ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. PARITY UNPINNED by
the reference itself (see oracle/orc_linalg.hpp).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librolo_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("rolo_oracle.cpp", "rolo_oracle_front.cpp", "rolo_oracle_backend.cpp", "rolo_oracle.h", "rolo_oracle_front.h",
                                             "rolo_oracle_backend.h", "orc_linalg.hpp", "orc_kdtree.hpp", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "librolo_oracle.so"])
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [("k_correspondences", C.c_int), ("regularization", C.c_int), ("neighbor_search", C.c_int),
                ("voxel_type", C.c_int), ("voxel_resolution", C.c_double), ("polar_resolution", C.c_double * 3),
                ("optimizer", C.c_int), ("max_iterations", C.c_int), ("rotation_epsilon", C.c_double),
                ("transformation_epsilon", C.c_double), ("lm_max_iterations", C.c_int),
                ("lm_init_lambda_factor", C.c_double), ("num_threads", C.c_int), ("fixed_iterations", C.c_int),
                ("q2_intended", C.c_int)]


class FrontParams(C.Structure):
    _fields_ = [("n_scan", C.c_int), ("horizon_scan", C.c_int), ("downsample_rate", C.c_int),
                ("lidar_min_range", C.c_float), ("lidar_max_range", C.c_float), ("edge_threshold", C.c_float),
                ("surf_threshold", C.c_float), ("odometry_surf_leaf_size", C.c_float)]


class LmScript(C.Structure):
    _fields_ = [("n_outer", C.c_int), ("n_trial", C.c_int), ("lin_y", C.POINTER(C.c_double)), ("lin_H", C.POINTER(C.c_double)),
                ("lin_b", C.POINTER(C.c_double)), ("lin_n", C.POINTER(C.c_int32)), ("err_y", C.POINTER(C.c_double))]


class TraceRec(C.Structure):
    _fields_ = [("stage", C.c_int), ("outer", C.c_int), ("trial", C.c_int), ("accepted", C.c_int),
                ("y0", C.c_double), ("yi", C.c_double), ("rho", C.c_double), ("lambda_", C.c_double),
                ("dnorm", C.c_double)]


REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS, REG_PLANE_S = range(6)
DIRECT27, DIRECT7, DIRECT1 = range(3)
VOXEL_POLAR, VOXEL_UNIFORM = range(2)
OPT_GN, OPT_LM, OPT_SO3_LM = range(3)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_reg_create.restype = C.c_void_p
        L.orc_reg_create.argtypes = [C.POINTER(Params)]
        L.orc_reg_destroy.argtypes = [C.c_void_p]
        for f in (L.orc_reg_set_target, L.orc_reg_set_source):
            f.argtypes = [C.c_void_p, fp, C.c_int, C.c_int]
        for f in (L.orc_reg_compute_covariances, L.orc_reg_build_voxelmap, L.orc_reg_num_voxels,
                  L.orc_reg_num_correspondences):
            f.argtypes = [C.c_void_p]
        for f in (L.orc_reg_get_source_covs, L.orc_reg_get_target_covs):
            f.argtypes = [C.c_void_p, dp]
        L.orc_reg_set_source_covs.argtypes = [C.c_void_p, dp]
        L.orc_reg_get_voxels.argtypes = [C.c_void_p, ip, ip, dp, dp]
        L.orc_reg_so3_linearize.restype = C.c_double
        L.orc_reg_so3_linearize.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_reg_linearize.restype = C.c_double
        L.orc_reg_linearize.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_reg_compute_error.restype = C.c_double
        L.orc_reg_compute_error.argtypes = [C.c_void_p, dp]
        L.orc_reg_get_correspondences.argtypes = [C.c_void_p, ip, ip, dp]
        L.orc_reg_t3_linearize.restype = C.c_double
        L.orc_reg_t3_linearize.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, C.c_double, C.c_float, dp, dp]
        L.orc_reg_compute_t_error.restype = C.c_double
        L.orc_reg_compute_t_error.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, C.c_double, C.c_float]
        L.orc_reg_align.argtypes = [C.c_void_p, fp, fp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_reg_compute_translation.argtypes = [C.c_void_p, dp, dp, dp, C.c_double, C.c_double, C.c_float,
                                                  C.POINTER(C.c_int)]
        L.orc_reg_trace.argtypes = [C.c_void_p, C.POINTER(TraceRec), C.c_int]
        L.orc_reg_clear_trace.argtypes = [C.c_void_p]
        L.orc_reg_set_script.argtypes = [C.c_void_p, C.POINTER(LmScript)]
        L.orc_reg_set_driver_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.orc_knn.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, ip, fp]
        L.orc_voxel_keys.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, ip]
        L.orc_so3_exp.argtypes = [dp, dp]
        L.orc_se3_exp.argtypes = [dp, dp, dp]
        L.orc_svd3.argtypes = [dp, dp, dp, dp]
        L.orc_ldlt_solve.argtypes = [C.c_int, dp, dp, dp]
        L.orc_transform_cloud_f.argtypes = [fp, fp, C.c_int, C.c_int, fp]
        L.orc_project.argtypes = [C.POINTER(FrontParams), fp, C.c_int, C.POINTER(C.c_uint16), C.c_int, fp, fp, fp,
                                  ip, fp, ip, ip]
        L.orc_extract_features.argtypes = [C.POINTER(FrontParams), fp, C.c_int, ip, fp, ip, ip, fp, ip, ip, fp, ip,
                                           fp, ip]
        L.orc_voxelgrid.argtypes = [fp, C.c_int, C.c_float, fp]
        L.orc_odom_create.restype = C.c_void_p
        L.orc_odom_create.argtypes = [C.POINTER(Params), C.c_float]
        L.orc_odom_destroy.argtypes = [C.c_void_p]
        L.orc_odom_backend_odometry.argtypes = [C.c_void_p, C.c_double]
        L.orc_odom_cloud.argtypes = [C.c_void_p, C.c_double, fp, C.c_int, fp, C.c_int, fp, dp, dp]
        L.orc_get_transformation.argtypes = [C.c_float] * 6 + [fp]
        L.orc_get_translation_and_euler.argtypes = [fp, fp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def default_params(**kw) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "polar_resolution":
            for j in range(3):
                p.polar_resolution[j] = float(v[j])
        else:
            setattr(p, k, v)
    return p


def front_params(**kw) -> FrontParams:
    p = FrontParams()
    lib().orc_front_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Reg:
    """Mirror of fast_gicp::RotVGICP<PointXYZI,PointXYZI> (rot_vgicp.hpp:72-104) on the oracle."""

    def __init__(self, params: Params | None = None):
        self.p = params if params is not None else default_params()
        self.h = lib().orc_reg_create(C.byref(self.p))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_reg_destroy(self.h)
            self.h = None

    def set_target(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        rc = lib().orc_reg_set_target(self.h, _f(pts), pts.shape[0], pts.shape[1])
        assert rc == 0, rc
        self.nt = pts.shape[0]

    def set_source(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        rc = lib().orc_reg_set_source(self.h, _f(pts), pts.shape[0], pts.shape[1])
        assert rc == 0, rc
        self.ns = pts.shape[0]

    def compute_covariances(self):
        return lib().orc_reg_compute_covariances(self.h)

    def source_covs(self):
        out = np.zeros((self.ns, 4, 4))
        assert lib().orc_reg_get_source_covs(self.h, _d(out)) == 0
        return out

    def set_source_covs(self, covs):
        a = np.ascontiguousarray(covs, np.float64).reshape(self.ns, 16)
        assert lib().orc_reg_set_source_covs(self.h, _d(a)) == 0

    def target_covs(self):
        out = np.zeros((self.nt, 4, 4))
        assert lib().orc_reg_get_target_covs(self.h, _d(out)) == 0
        return out

    def build_voxelmap(self):
        return lib().orc_reg_build_voxelmap(self.h)

    def voxels(self):
        V = lib().orc_reg_num_voxels(self.h)
        assert V >= 0
        keys = np.zeros((V, 3), np.int32); counts = np.zeros(V, np.int32)
        means = np.zeros((V, 4)); covs = np.zeros((V, 4, 4))
        lib().orc_reg_get_voxels(self.h, _i(keys), _i(counts), _d(means), _d(covs))
        return keys, counts, means, covs

    def so3_linearize(self, T):
        T = np.ascontiguousarray(T, np.float64); H = np.zeros((3, 3)); b = np.zeros(3)
        e = lib().orc_reg_so3_linearize(self.h, _d(T), _d(H), _d(b))
        return e, H, b

    def linearize(self, T):
        T = np.ascontiguousarray(T, np.float64); H = np.zeros((6, 6)); b = np.zeros(6)
        e = lib().orc_reg_linearize(self.h, _d(T), _d(H), _d(b))
        return e, H, b

    def compute_error(self, T):
        T = np.ascontiguousarray(T, np.float64)
        return lib().orc_reg_compute_error(self.h, _d(T))

    def correspondences(self, with_mahalanobis=False):
        n = lib().orc_reg_num_correspondences(self.h)
        s = np.zeros(n, np.int32); v = np.zeros(n, np.int32)
        m = np.zeros((n, 4, 4)) if with_mahalanobis else None
        lib().orc_reg_get_correspondences(self.h, _i(s), _i(v), _d(m))
        return (s, v, m) if with_mahalanobis else (s, v)

    def t3_linearize(self, t, g, l, dtn=0.1, dtn1=0.1, ct_lambda=0.3):
        t, g, l = (np.ascontiguousarray(a, np.float64) for a in (t, g, l))
        H = np.zeros((6, 6)); b = np.zeros(6)
        e = lib().orc_reg_t3_linearize(self.h, _d(t), _d(g), _d(l), dtn, dtn1, ct_lambda, _d(H), _d(b))
        return e, H, b

    def compute_t_error(self, t, g, l, dtn=0.1, dtn1=0.1, ct_lambda=0.3):
        t, g, l = (np.ascontiguousarray(a, np.float64) for a in (t, g, l))
        return lib().orc_reg_compute_t_error(self.h, _d(t), _d(g), _d(l), dtn, dtn1, ct_lambda)

    def align(self, guess=None):
        g = np.ascontiguousarray(guess, np.float32) if guess is not None else None
        Tf = np.zeros((4, 4), np.float32); Td = np.zeros((4, 4))
        it = C.c_int(0); cv = C.c_int(0)
        rc = lib().orc_reg_align(self.h, _f(g), _f(Tf), _d(Td), C.byref(it), C.byref(cv))
        return rc, Tf, Td, it.value, bool(cv.value)

    def compute_translation(self, trans, init_guess, last_t0, dtn=0.1, dtn1=0.1, ct_lambda=0.3):
        t = np.array(trans, np.float64)
        g = np.ascontiguousarray(init_guess, np.float64); l = np.ascontiguousarray(last_t0, np.float64)
        it = C.c_int(0)
        rc = lib().orc_reg_compute_translation(self.h, _d(t), _d(g), _d(l), dtn, dtn1, ct_lambda, C.byref(it))
        return rc, t, it.value

    def trace(self):
        n = lib().orc_reg_trace(self.h, None, 0)
        arr = (TraceRec * max(n, 1))()
        lib().orc_reg_trace(self.h, arr, n)
        return [dict(stage=a.stage, outer=a.outer, trial=a.trial, accepted=a.accepted, y0=a.y0, yi=a.yi, rho=a.rho,
                     lam=a.lambda_, dnorm=a.dnorm) for a in arr[:n]]

    def clear_trace(self):
        lib().orc_reg_clear_trace(self.h)

    def set_driver_params(self, **kw):
        """the LM drivers' knobs on the live object (caches kept): optimizer, max_iterations, rotation_epsilon, transformation_epsilon, lm_max_iterations, lm_init_lambda_factor, fixed_iterations"""
        for k, v in kw.items():
            setattr(self.p, k, v)
        lib().orc_reg_set_driver_params(self.h, C.byref(self.p))

    def set_script(self, lin_y=None, lin_H=None, lin_b=None, lin_n=None, err_y=None):
        """scripted evaluations (orc_reg_set_script): lin_y [O], lin_H [O, 6, 6], lin_b [O, 6], lin_n [O], err_y [O, T]; no arguments: back to real evaluations"""
        if lin_y is None:
            lib().orc_reg_set_script(self.h, None); self._script = None
            return
        a = dict(lin_y=np.ascontiguousarray(lin_y, np.float64), lin_H=np.ascontiguousarray(lin_H, np.float64), lin_b=np.ascontiguousarray(lin_b, np.float64),
                 lin_n=np.ascontiguousarray(lin_n, np.int32), err_y=np.ascontiguousarray(err_y, np.float64))
        O, T = a["err_y"].shape
        assert a["lin_y"].shape == (O,) and a["lin_H"].shape == (O, 6, 6) and a["lin_b"].shape == (O, 6) and a["lin_n"].shape == (O,)
        sc = LmScript(O, T, _d(a["lin_y"]), _d(a["lin_H"]), _d(a["lin_b"]), _i(a["lin_n"]), _d(a["err_y"]))
        self._script = (sc, a)   # the oracle keeps pointers into these
        lib().orc_reg_set_script(self.h, C.byref(sc))


def knn(pts, k=20, threads=0):
    pts = np.ascontiguousarray(pts, np.float32)
    n = pts.shape[0]
    idx = np.zeros((n, k), np.int32); d2 = np.zeros((n, k), np.float32)
    rc = lib().orc_knn(_f(pts), n, pts.shape[1], k, threads, _i(idx), _f(d2))
    assert rc == 0, rc
    return idx, d2


def voxel_keys(pts, voxel_type, voxel_resolution=1.0, polar_res=(0.175, 0.175, 2.0), T=None):
    pts = np.ascontiguousarray(pts, np.float32)
    keys = np.zeros((pts.shape[0], 3), np.int32)
    pr = np.ascontiguousarray(polar_res, np.float64)
    Tm = np.ascontiguousarray(T, np.float64) if T is not None else None
    lib().orc_voxel_keys(_f(pts), pts.shape[0], pts.shape[1], voxel_type, voxel_resolution, _d(pr), _d(Tm), _i(keys))
    return keys


def so3_exp(w):
    w = np.ascontiguousarray(w, np.float64); R = np.zeros((3, 3))
    lib().orc_so3_exp(_d(w), _d(R))
    return R


def se3_exp(a):
    a = np.ascontiguousarray(a, np.float64); R = np.zeros((3, 3)); t = np.zeros(3)
    lib().orc_se3_exp(_d(a), _d(R), _d(t))
    return R, t


def svd3(A):
    A = np.ascontiguousarray(A, np.float64); U = np.zeros((3, 3)); s = np.zeros(3); V = np.zeros((3, 3))
    lib().orc_svd3(_d(A), _d(U), _d(s), _d(V))
    return U, s, V


def ldlt_solve(A, b):
    A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64); x = np.zeros_like(b)
    rc = lib().orc_ldlt_solve(A.shape[0], _d(A), _d(b), _d(x))
    return rc, x


def transform_cloud_f(pts, T):
    pts = np.ascontiguousarray(pts, np.float32); T = np.ascontiguousarray(T, np.float32)
    out = np.zeros_like(pts)
    lib().orc_transform_cloud_f(_f(pts), _f(out), pts.shape[0], pts.shape[1], _f(T))
    return out


class Deskew(C.Structure):
    _fields_ = [("enabled", C.c_int), ("odom_incre_rpy", C.c_float * 3), ("scan_period", C.c_float), ("odom_time_diff", C.c_double)]


def deskew(incre_rpy, scan_period=0.1, odom_time_diff=0.1, enabled=1):
    d = Deskew(); d.enabled = enabled; d.scan_period = scan_period; d.odom_time_diff = odom_time_diff
    for k in range(3):
        d.odom_incre_rpy[k] = incre_rpy[k]
    return d


def azimuth_times(xyz, scan_period=0.1):
    a = np.ascontiguousarray(xyz, np.float32); out = np.zeros(a.shape[0], np.float32)
    f = lib().orc_azimuth_times
    f.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]; f.restype = None
    f(_f(a), a.shape[1], a.shape[0], scan_period, _f(out))
    return out


def odom_increment(front6, back6):
    f = np.ascontiguousarray(front6, np.float32); b = np.ascontiguousarray(back6, np.float32); o = np.zeros(6, np.float32)
    lib().orc_odom_increment(_f(f), _f(b), _f(o))
    return o


def project(fp: FrontParams, xyz, ring, rel_time=None, dsk=None):
    xyz = np.ascontiguousarray(xyz, np.float32); ring = np.ascontiguousarray(ring, np.uint16)
    NS, H = fp.n_scan, fp.horizon_scan
    range_mat = np.zeros(NS * H, np.float32); full = np.zeros((NS * H, 4), np.float32)
    ext = np.zeros((NS * H, 4), np.float32); col = np.zeros(NS * H, np.int32); rng = np.zeros(NS * H, np.float32)
    sr = np.zeros(NS, np.int32); er = np.zeros(NS, np.int32)
    rt = np.ascontiguousarray(rel_time, np.float32) if rel_time is not None else None
    lib().orc_project_deskew.restype = C.c_int
    n = lib().orc_project_deskew(C.byref(fp), _f(xyz), xyz.shape[1], ring.ctypes.data_as(C.POINTER(C.c_uint16)),
                                 xyz.shape[0], _f(rt) if rt is not None else None, C.byref(dsk) if dsk is not None else None,
                                 _f(range_mat), _f(full), _f(ext), _i(col), _f(rng), _i(sr), _i(er))
    assert n >= 0
    return dict(n=n, range_mat=range_mat.reshape(NS, H), full_cloud=full, extracted=ext[:n].copy(),
                point_col_ind=col[:n].copy(), point_range=rng[:n].copy(), start_ring=sr, end_ring=er)


def extract_features(fp: FrontParams, proj):
    n = proj["n"]
    ext = np.ascontiguousarray(proj["extracted"], np.float32)
    col = np.ascontiguousarray(proj["point_col_ind"], np.int32); rng = np.ascontiguousarray(proj["point_range"], np.float32)
    sr = np.ascontiguousarray(proj["start_ring"], np.int32); er = np.ascontiguousarray(proj["end_ring"], np.int32)
    curv = np.zeros(n, np.float32); picked = np.zeros(n, np.int32); label = np.zeros(n, np.int32)
    corner = np.zeros((max(n, 1), 4), np.float32); surf = np.zeros((max(n, 1), 4), np.float32)
    nc = C.c_int32(0); ns = C.c_int32(0)
    rc = lib().orc_extract_features(C.byref(fp), _f(ext), n, _i(col), _f(rng), _i(sr), _i(er), _f(curv), _i(picked),
                                    _i(label), _f(corner), C.byref(nc), _f(surf), C.byref(ns))
    assert rc == 0
    return dict(curvature=curv, picked=picked, label=label, corner=corner[:nc.value].copy(),
                surface=surf[:ns.value].copy())


def voxelgrid(pts, leaf):
    pts = np.ascontiguousarray(pts, np.float32)
    out = np.zeros_like(pts)
    f = lib().orc_voxelgrid
    f.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.POINTER(C.c_float)]; f.restype = C.c_int
    m = f(_f(pts), pts.shape[0], leaf, _f(out))
    return out[:m].copy()


def get_transformation(x, y, z, roll, pitch, yaw):
    T = np.zeros(16, np.float32)
    f = lib().orc_get_transformation
    f.argtypes = [C.c_float] * 6 + [C.POINTER(C.c_float)]; f.restype = None
    f(x, y, z, roll, pitch, yaw, _f(T))
    return T


def affine3f_rotation(T):
    """Eigen::Affine3f::rotation() of a 4x4 (or 3x3) float matrix"""
    A = np.eye(4, dtype=np.float32); T = np.asarray(T, np.float32); A[:T.shape[0], :T.shape[1]] = T
    R = np.zeros((3, 3), np.float32)
    f = lib().orc_affine3f_rotation
    f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]; f.restype = None
    f(_f(A), _f(R))
    return R


def get_translation_and_euler(T16):
    T = np.ascontiguousarray(T16, np.float32).reshape(16); o = np.zeros(6, np.float32)
    f = lib().orc_get_translation_and_euler
    f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]; f.restype = None
    f(_f(T), _f(o))
    return o


class Odom:
    def __init__(self, reg_params: Params, ct_lambda=0.3):
        self.p = reg_params
        self.h = lib().orc_odom_create(C.byref(self.p), ct_lambda)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_odom_destroy(self.h)
            self.h = None

    def backend_odometry(self, stamp):
        lib().orc_odom_backend_odometry(self.h, stamp)

    def cloud(self, stamp, corner, surface):
        corner = np.ascontiguousarray(corner, np.float32).reshape(-1, 4)
        surface = np.ascontiguousarray(surface, np.float32).reshape(-1, 4)
        pose = np.zeros(6, np.float32); R = np.zeros((3, 3)); t = np.zeros(3)
        rc = lib().orc_odom_cloud(self.h, stamp, _f(corner), corner.shape[0], _f(surface), surface.shape[0],
                                  _f(pose), _d(R), _d(t))
        return rc, pose, R, t


def scan2map(corner, surf, map_corner, map_surf, tf6, edge_min=10, surf_min=100, threads=0):
    """orc_scan2map (src/backMapping.cpp:681-1058). Returns (transformTobeMapped, stats dict, selected flags, coeffs of the last iteration)."""
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_scan2map.restype = C.c_int
    L.orc_scan2map.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_ubyte), fp]
    a = [np.ascontiguousarray(x, np.float32) for x in (corner, surf, map_corner, map_surf)]
    assert all(x.ndim == 2 and x.shape[1] == 4 for x in a)
    tf = np.ascontiguousarray(tf6, np.float32).copy()
    n = a[0].shape[0] + a[1].shape[0]
    st = (C.c_int * 5)(); sel = np.zeros(n, np.uint8); co = np.zeros((n, 4), np.float32)
    rc = L.orc_scan2map(_f(a[0]), a[0].shape[0], _f(a[1]), a[1].shape[0], _f(a[2]), a[2].shape[0], _f(a[3]), a[3].shape[0], _f(tf), edge_min, surf_min, threads,
                        st, sel.ctypes.data_as(C.POINTER(C.c_ubyte)), _f(co))
    assert rc == 0
    return tf, dict(zip(("skipped", "iterations", "converged", "degenerate", "n_selected"), [int(v) for v in st])), sel.astype(bool), co
