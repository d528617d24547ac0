"""Host-side mirror of the reference operator `fast_gicp::RotVGICP<pcl::PointXYZI, pcl::PointXYZI>`
(reference include/rot_gicp/gicp/rot_vgicp.hpp:72-104, lsq_registration.hpp:51-62) over the C ABI of
librolo_hip.so. Same method names, argument meaning and error behaviour as the C++ class; clouds are (n, >=3)
float32 arrays (x, y, z first — pcl::PointXYZI is 8 floats per point). Plumbing only: all compute is in the HIP
library. The C++ twin of this class is include/rot_vgicp_hip.hpp.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from . import _lib
from ._lib import Params, Stats, TraceRec, check, lib

# gicp_settings.hpp:6-13, lsq_registration.hpp:13
class RegularizationMethod:
    NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS, PLANE_S = range(6)


class NeighborSearchMethod:
    DIRECT27, DIRECT7, DIRECT1 = range(3)


class VoxelType:
    POLAR, UNIFORM = range(2)


class LSQ_OPTIMIZER_TYPE:
    GaussNewton, LevenbergMarquardt, SO3_LevenbergMarquardt = range(3)


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


class RotVGICP:
    def __init__(self, device: int = 0, _borrowed=None):
        if _borrowed is not None:   # member of a RotVGICPBatch: the batch owns the context
            h = C.c_void_p(_borrowed)
        else:
            h = C.c_void_p()
            check(lib().rolo_ctx_create(device, C.byref(h)), "rolo_ctx_create")
        self._owned = _borrowed is None
        self._h = h
        self._p = Params()
        lib().rolo_default_params(C.byref(self._p))
        self._ns = self._nt = 0
        self._src_ref = self._tgt_ref = None
        self.last_stats = None
        self.last_translation_stats = None

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                lib().rolo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setters (rot_vgicp_impl.hpp:44-99, lsq_registration_impl.hpp:29-42,337-340) ----
    def _push(self):
        check(lib().rolo_set_params(self._h, C.byref(self._p)), "rolo_set_params")

    def setResolution(self, resolution: float):
        self._p.voxel_resolution = resolution
        self._p.voxel_type = VoxelType.UNIFORM
        self._push()

    def setPolarResolution(self, theta_res: float, phi_res: float, r_res: float):
        self._p.polar_resolution[0], self._p.polar_resolution[1], self._p.polar_resolution[2] = theta_res, phi_res, r_res
        self._p.voxel_type = VoxelType.POLAR
        self._push()

    def setNumThreads(self, n: int):
        """Accepted for API compatibility; the HIP path has no host threads to size."""

    def setCorrespondenceRandomness(self, k: int):
        self._p.k_correspondences = k
        self._push()

    def setRegularizationMethod(self, method: int):
        self._p.regularization = method
        self._push()

    def setNeighborSearchMethod(self, method: int):
        self._p.neighbor_search = method
        self._push()

    def setVoxelAccumulationMode(self, mode: int):
        """ADDITIVE, ADDITIVE_WEIGHTED and MULTIPLICATIVE all build AdditiveVmfVoxel in the reference
        (vmp_voxel.hpp:176-184), so the mode has no effect there either."""

    def setOptimizerType(self, t: int):
        self._p.optimizer = t
        self._push()

    def setRotationEpsilon(self, eps: float):
        self._p.rotation_epsilon = eps
        self._push()

    def setTransformationEpsilon(self, eps: float):
        self._p.transformation_epsilon = eps
        self._push()

    def setMaximumIterations(self, n: int):
        self._p.max_iterations = n
        self._push()

    def setInitialLambdaFactor(self, f: float):
        self._p.lm_init_lambda_factor = f
        self._push()

    def setLmMaxIterations(self, n: int):
        """lm_max_iterations_: a protected member without a setter in the reference (lsq_registration.hpp:104, default 10) — what a subclass can reach"""
        self._p.lm_max_iterations = n
        self._push()

    def setFixedIterations(self, n: int):
        """Harness knob (not in the reference): run exactly n outer iterations in align()."""
        self._p.fixed_iterations = n
        self._push()

    def setOverlapKnn(self, on: bool):
        """Tuning knob (not in the reference): search source and target in one chain of launches (default on)."""
        self._p.overlap_knn = int(on)
        self._push()

    def setUseGraph(self, on: bool):
        """Tuning knob (not in the reference): hipGraph capture / replay of the per-frame schedule (default on)."""
        self._p.use_graph = int(on)
        self._push()

    def setLoadHint(self, mode: int):
        """-1 (default): register_async picks its kernels per frame from the device's load; 0 / 1 pin the idle- / busy-device choice (rolo_hip.h)"""
        check(lib().rolo_set_load_hint(self._h, int(mode)), "rolo_set_load_hint")

    def setFusedLm(self, on):
        """tuning knob: 0 / False = pass + controller launches, 1 / True = one launch per LM trial (controller in the prologue of the next pass), 2 = one launch per
        frame (the resident LM kernel); see rolo_hip.h"""
        self._p.fused_lm = int(on)
        self._push()

    def setQ2Intended(self, on: bool):
        self._p.q2_intended = int(on)
        self._push()

    # ---- clouds ----
    def setInputTarget(self, cloud: np.ndarray):
        if cloud is self._tgt_ref:  # same pointer => no-op (rot_vgicp_impl.hpp:134-136)
            return
        a = np.ascontiguousarray(cloud, np.float32)
        check(lib().rolo_set_target(self._h, _f(a), a.shape[0], a.shape[1]), "rolo_set_target")
        self._tgt_ref, self._nt = cloud, a.shape[0]

    def setInputSource(self, cloud: np.ndarray):
        if cloud is self._src_ref:
            return
        a = np.ascontiguousarray(cloud, np.float32)
        check(lib().rolo_set_source(self._h, _f(a), a.shape[0], a.shape[1]), "rolo_set_source")
        self._src_ref, self._ns = cloud, a.shape[0]

    def setInputTargetDevice(self, dev_ptr: int, n: int, stride: int):
        check(lib().rolo_set_target_device(self._h, C.c_void_p(dev_ptr), n, stride), "rolo_set_target_device")
        self._tgt_ref, self._nt = None, n

    def setInputSourceDevice(self, dev_ptr: int, n: int, stride: int):
        check(lib().rolo_set_source_device(self._h, C.c_void_p(dev_ptr), n, stride), "rolo_set_source_device")
        self._src_ref, self._ns = None, n

    def clearSource(self):
        check(lib().rolo_clear_source(self._h), "rolo_clear_source")
        self._src_ref, self._ns = None, 0

    def clearTarget(self):
        check(lib().rolo_clear_target(self._h), "rolo_clear_target")
        self._tgt_ref, self._nt = None, 0

    def swapSourceAndTarget(self):
        check(lib().rolo_swap_source_and_target(self._h), "rolo_swap_source_and_target")
        self._src_ref, self._tgt_ref = self._tgt_ref, self._src_ref
        self._ns, self._nt = self._nt, self._ns

    # ---- covariances ----
    def adoptTargetCovariances(self):
        """The source is the previous target moved by a pure translation: take over its covariances (not a reference
        call; see rolo_adopt_target_covariances). After setInputSource*, before setInputTarget*."""
        check(lib().rolo_adopt_target_covariances(self._h), "rolo_adopt_target_covariances")

    def computeCovariances(self):
        check(lib().rolo_compute_covariances(self._h), "rolo_compute_covariances")

    def getSourceCovariances(self):
        out = np.zeros((self._ns, 4, 4))
        check(lib().rolo_get_source_covariances(self._h, _d(out)), "rolo_get_source_covariances")
        return out

    def getTargetCovariances(self):
        out = np.zeros((self._nt, 4, 4))
        check(lib().rolo_get_target_covariances(self._h, _d(out)), "rolo_get_target_covariances")
        return out

    def setSourceCovariances(self, covs):
        a = np.ascontiguousarray(covs, np.float64).reshape(self._ns, 16)
        check(lib().rolo_set_source_covariances(self._h, _d(a)), "rolo_set_source_covariances")

    def setTargetCovariances(self, covs):
        a = np.ascontiguousarray(covs, np.float64).reshape(self._nt, 16)
        check(lib().rolo_set_target_covariances(self._h, _d(a)), "rolo_set_target_covariances")

    def knn(self, which: int):
        n = self._ns if which == 0 else self._nt
        k = self._p.k_correspondences
        idx = np.zeros((n, k), np.int32); d2 = np.zeros((n, k), np.float32)
        check(lib().rolo_get_knn(self._h, which, _i(idx), _f(d2)), "rolo_get_knn")
        return idx, d2

    # ---- voxel map ----
    def buildVoxelMap(self):
        check(lib().rolo_build_voxelmap(self._h), "rolo_build_voxelmap")

    def numEdgePoints(self) -> int:
        """target points of the last map build within 1e-12 of a POLAR bin edge (rolo_num_edge_points)"""
        return check(lib().rolo_num_edge_points(self._h), "rolo_num_edge_points")

    def voxels(self):
        V = check(lib().rolo_num_voxels(self._h), "rolo_num_voxels")
        keys = np.zeros((V, 3), np.int32); counts = np.zeros(V, np.int32); means = np.zeros((V, 4)); covs = np.zeros((V, 4, 4))
        check(lib().rolo_get_voxels(self._h, _i(keys), _i(counts), _d(means), _d(covs)), "rolo_get_voxels")
        return keys, counts, means, covs

    def targetVoxelKeys(self):
        keys = np.zeros((self._nt, 3), np.int32)
        check(lib().rolo_get_target_voxel_keys(self._h, _i(keys)), "rolo_get_target_voxel_keys")
        return keys

    # ---- stage-level evaluations ----
    def so3_linearize(self, T):
        T = np.ascontiguousarray(T, np.float64); H = np.zeros((3, 3)); b = np.zeros(3); e = C.c_double()
        check(lib().rolo_so3_linearize(self._h, _d(T), _d(H), _d(b), C.byref(e)), "rolo_so3_linearize")
        return e.value, H, b

    def linearize(self, T):
        T = np.ascontiguousarray(T, np.float64); H = np.zeros((6, 6)); b = np.zeros(6); e = C.c_double()
        check(lib().rolo_linearize(self._h, _d(T), _d(H), _d(b), C.byref(e)), "rolo_linearize")
        return e.value, H, b

    def compute_error(self, T):
        T = np.ascontiguousarray(T, np.float64); e = C.c_double()
        check(lib().rolo_compute_error(self._h, _d(T), C.byref(e)), "rolo_compute_error")
        return e.value

    def evaluateCost(self, relative_pose):
        """lsq_registration_impl.hpp:50-52: linearize at Isometry3f(relative_pose).cast<double>()."""
        return self.linearize(np.asarray(relative_pose, np.float32).astype(np.float64))

    def correspondences(self):
        noff = {NeighborSearchMethod.DIRECT1: 1, NeighborSearchMethod.DIRECT7: 7, NeighborSearchMethod.DIRECT27: 27}[self._p.neighbor_search]
        found = np.zeros((self._ns, noff), np.int32); keys = np.zeros((self._ns, noff, 3), np.int32)
        check(lib().rolo_get_correspondences(self._h, _i(found), _i(keys)), "rolo_get_correspondences")
        return found, keys

    def t3_linearize(self, t, g, l, dtn=0.1, dtn1=0.1, ct_lambda=0.3):
        t, g, l = (np.ascontiguousarray(a, np.float64) for a in (t, g, l))
        H = np.zeros((6, 6)); b = np.zeros(6); e = C.c_double()
        check(lib().rolo_t3_linearize(self._h, _d(t), _d(g), _d(l), dtn, dtn1, ct_lambda, _d(H), _d(b), C.byref(e)), "rolo_t3_linearize")
        return e.value, H, b

    def compute_t_error(self, t, g, l, dtn=0.1, dtn1=0.1, ct_lambda=0.3):
        t, g, l = (np.ascontiguousarray(a, np.float64) for a in (t, g, l))
        e = C.c_double()
        check(lib().rolo_compute_t_error(self._h, _d(t), _d(g), _d(l), dtn, dtn1, ct_lambda, C.byref(e)), "rolo_compute_t_error")
        return e.value

    # ---- drivers ----
    def align(self, guess=None):
        """pcl::Registration::align. Returns getFinalTransformation() (4x4 float32); the double pose is in
        self.final_transformation_d; hasConverged() / iteration count in self.last_stats."""
        g = np.ascontiguousarray(guess, np.float32) if guess is not None else None
        Tf = np.zeros((4, 4), np.float32); Td = np.zeros((4, 4)); st = Stats()
        check(lib().rolo_align(self._h, _f(g), _f(Tf), _d(Td), C.byref(st)), "rolo_align")
        self.final_transformation_d = Td
        self.last_stats = st
        return Tf

    def getFinalTransformation(self):
        return self.final_transformation_d.astype(np.float32)

    def hasConverged(self):
        return bool(self.last_stats.converged)

    def computeTranslation(self, trans, init_guess, last_t0, interval_tn=0.1, interval_tn_1=0.1, ct_lambda=0.3):
        """RotVGICP::computeTranslation: returns the optimised translation (the reference updates `trans` in place)."""
        t = np.array(trans, np.float64)
        g = np.ascontiguousarray(init_guess, np.float64); l = np.ascontiguousarray(last_t0, np.float64)
        st = Stats()
        check(lib().rolo_compute_translation(self._h, _d(t), _d(g), _d(l), interval_tn, interval_tn_1, ct_lambda, C.byref(st)), "rolo_compute_translation")
        self.last_translation_stats = st
        return t

    def register_async(self, guess, trans_start, init_guess, last_t0, interval_tn=0.1, interval_tn_1=0.1, ct_lambda=0.3):
        g = np.ascontiguousarray(guess, np.float32) if guess is not None else None
        ts = np.ascontiguousarray(trans_start, np.float64); ig = np.ascontiguousarray(init_guess, np.float64); l0 = np.ascontiguousarray(last_t0, np.float64)
        check(lib().rolo_register_async(self._h, _f(g), _d(ts), _d(ig), _d(l0), interval_tn, interval_tn_1, ct_lambda), "rolo_register_async")

    def register_wait(self):
        Tf = np.zeros((4, 4), np.float32); Td = np.zeros((4, 4)); t = np.zeros(3); rs = Stats(); ts = Stats()
        check(lib().rolo_register_wait(self._h, _f(Tf), _d(Td), _d(t), C.byref(rs), C.byref(ts)), "rolo_register_wait")
        self.final_transformation_d = Td; self.last_stats = rs; self.last_translation_stats = ts
        return Tf, Td, t

    # ---- test hook: the LM controllers on scripted pass results (rolo_debug_lm_script_*) ----
    @staticmethod
    def _script(lin_y, lin_H, lin_b, lin_n, err_y):
        from ._lib import LmScript
        a = dict(lin_y=np.ascontiguousarray(lin_y, np.float64), lin_H=np.ascontiguousarray(lin_H, np.float64), lin_b=np.ascontiguousarray(lin_b, np.float64),
                 lin_n=np.ascontiguousarray(lin_n, np.int32), err_y=np.ascontiguousarray(err_y, np.float64))
        O, T = a["err_y"].shape
        assert a["lin_y"].shape == (O,) and a["lin_H"].shape == (O, 6, 6) and a["lin_b"].shape == (O, 6) and a["lin_n"].shape == (O,)
        return LmScript(O, T, _d(a["lin_y"]), _d(a["lin_H"]), _d(a["lin_b"]), _i(a["lin_n"]), _d(a["err_y"])), a

    def script_align(self, script, guess=None, generic_ctrl=False):
        """computeTransformation on scripted evaluations (script = the five arrays of rolo_lm_script); returns the error code (0, or ROLO_ENOCORR, ...)"""
        sc, keep = self._script(*script)
        g = np.ascontiguousarray(guess, np.float32) if guess is not None else None
        Tf = np.zeros((4, 4), np.float32); Td = np.zeros((4, 4)); st = Stats()
        rc = lib().rolo_debug_lm_script_align(self._h, C.byref(sc), _f(g), int(generic_ctrl), _f(Tf), _d(Td), C.byref(st))
        self.final_transformation_d = Td; self.last_stats = st
        return rc

    def script_translation(self, script, trans, init_guess, last_t0, interval_tn=0.1, interval_tn_1=0.1, ct_lambda=0.3, generic_ctrl=False):
        sc, keep = self._script(*script)
        t = np.array(trans, np.float64)
        g = np.ascontiguousarray(init_guess, np.float64); l = np.ascontiguousarray(last_t0, np.float64)
        st = Stats()
        rc = lib().rolo_debug_lm_script_translation(self._h, C.byref(sc), _d(t), _d(g), _d(l), interval_tn, interval_tn_1, ct_lambda, int(generic_ctrl), C.byref(st))
        self.last_translation_stats = st
        return rc, t

    def getFinalHessian(self):
        H = np.zeros((6, 6))
        check(lib().rolo_get_final_hessian(self._h, _d(H)), "rolo_get_final_hessian")
        return H

    def trace(self):
        n = check(lib().rolo_get_trace(self._h, None, 0), "rolo_get_trace")
        arr = (TraceRec * max(n, 1))()
        check(lib().rolo_get_trace(self._h, arr, n), "rolo_get_trace")
        return [dict(stage=a.stage, outer=a.outer, trial=a.trial, accepted=a.accepted, y0=a.y0, yi=a.yi, rho=a.rho,
                     lam=a.lambda_, dnorm=a.dnorm) for a in arr[:n]]

    def transformPointCloud(self, cloud, T):
        a = np.ascontiguousarray(cloud, np.float32); Tm = np.ascontiguousarray(T, np.float32)
        out = np.zeros_like(a)
        check(lib().rolo_transform_cloud(self._h, _f(a), _f(out), a.shape[0], a.shape[1], _f(Tm)), "rolo_transform_cloud")
        return out

    def counters(self) -> dict:
        """rolo_ctx_counters as a dict"""
        v = (C.c_longlong * 14)()
        check(lib().rolo_ctx_counters(self._h, v, 14), "rolo_ctx_counters")
        return dict(zip(("frames", "graph_replays", "graph_captures", "eager_frames", "topup_frames", "sync_chunks", "hint_rot", "hint_trans", "walk_lanes",
                         "host_enqueue_ns", "host_wait_blocked_ns", "host_wait_other_ns", "persist_bails", "load_mode"), [int(x) for x in v]))

    def debug_chain(self, kind: int, n_pairs: int, grid: int, reps: int):
        """experiment hook (rolo_debug_chain): `reps` replays of a captured chain of launch pairs on this context's stream; asynchronous"""
        check(lib().rolo_debug_chain(self._h, kind, n_pairs, grid, reps), "rolo_debug_chain")

    def synchronize(self):
        import ctypes as _C
        hip = _C.CDLL("libamdhip64.so", mode=_C.RTLD_GLOBAL) if not hasattr(RotVGICP, "_hip") else RotVGICP._hip
        RotVGICP._hip = hip
        hip.hipStreamSynchronize.argtypes = [_C.c_void_p]
        return hip.hipStreamSynchronize(_C.c_void_p(self.stream))

    @property
    def stream(self) -> int:
        return int(lib().rolo_ctx_stream(self._h) or 0)

    # ---- multi-GPU ----
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(lib().rolo_comm_unique_id(buf), "rolo_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = C.create_string_buffer(unique_id, 128)
        check(lib().rolo_comm_init(self._h, buf, rank, world), "rolo_comm_init")

    def comm_info(self):
        """(rank, world) read back from the RCCL communicator; world 0 = none"""
        r, w = C.c_int(), C.c_int()
        check(lib().rolo_comm_info(self._h, C.byref(r), C.byref(w)), "rolo_comm_info")
        return r.value, w.value

    # peer exchange (rolo_peer_*): the sharded path without a collective library — export a mailbox, swap the 64-byte handles, connect
    def peer_export(self, world: int, max_points: int) -> bytes:
        buf = C.create_string_buffer(64)
        check(lib().rolo_peer_export(self._h, world, int(max_points), buf), "rolo_peer_export")
        return buf.raw

    def peer_connect(self, handles, rank: int, world: int):
        blob = b"".join(bytes(h) for h in handles)
        assert len(blob) == 64 * world
        check(lib().rolo_peer_connect(self._h, C.create_string_buffer(blob, len(blob)), rank, world), "rolo_peer_connect")

    def peer_disconnect(self):
        check(lib().rolo_peer_disconnect(self._h), "rolo_peer_disconnect")

    def peer_selftest(self, reps: int = 8):
        """collective (every rank, same reps): known words through both exchanges, verified; returns (us per LM exchange, us of the covariance exchange)"""
        us = (C.c_double * 2)()
        check(lib().rolo_peer_selftest(self._h, int(reps), us), "rolo_peer_selftest")
        return float(us[0]), float(us[1])

    def peer_info(self):
        """(rank, world, memory kind of the mailbox); world 0 = not connected"""
        r, w = C.c_int(), C.c_int(); k = C.create_string_buffer(16)
        check(lib().rolo_peer_info(self._h, C.byref(r), C.byref(w), k), "rolo_peer_info")
        return r.value, w.value, k.value.decode()


class RotVGICPBatch:
    """B independent scan pairs registered with one call (BASELINE config 5; the reference would loop
    LidarOdometry::scanRegeistration, lidarOdometry.cpp:256-317, over them). `members[i]` is an ordinary RotVGICP
    — set its parameters and clouds as usual — owned by the batch."""

    def __init__(self, n_members: int, device: int = 0):
        h = C.c_void_p()
        check(lib().rolo_batch_create(device, n_members, C.byref(h)), "rolo_batch_create")
        self._h = h
        self.members = [RotVGICP(_borrowed=lib().rolo_batch_member(h, i)) for i in range(n_members)]

    def __len__(self):
        return len(self.members)

    def close(self):
        if getattr(self, "_h", None):
            for m in self.members:
                m.close()
            lib().rolo_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def register_async(self, guesses, trans_starts, init_guesses, last_t0s, interval_tn=0.1, interval_tn_1=0.1, ct_lambda=0.3):
        n = len(self.members)
        g = np.ascontiguousarray(guesses, np.float32).reshape(n, 16) if guesses is not None else None
        ts = np.ascontiguousarray(trans_starts, np.float64).reshape(n, 3) if trans_starts is not None else None
        ig = np.ascontiguousarray(init_guesses, np.float64).reshape(n, 3)
        l0 = np.ascontiguousarray(last_t0s, np.float64).reshape(n, 3)
        check(lib().rolo_batch_register_async(self._h, _f(g), _d(ts), _d(ig), _d(l0), interval_tn, interval_tn_1, ct_lambda),
              "rolo_batch_register_async")

    def register_wait(self):
        n = len(self.members)
        Tf = np.zeros((n, 4, 4), np.float32); Td = np.zeros((n, 4, 4)); t = np.zeros((n, 3))
        rs = (Stats * n)(); ts = (Stats * n)()
        check(lib().rolo_batch_register_wait(self._h, _f(Tf), _d(Td), _d(t), rs, ts), "rolo_batch_register_wait")
        for i, m in enumerate(self.members):
            m.final_transformation_d = Td[i]; m.last_stats = rs[i]; m.last_translation_stats = ts[i]
        return Tf, Td, t
