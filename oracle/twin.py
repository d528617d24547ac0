"""Independent numpy/scipy twin of the registration path — ORACLE-SIDE TEST INFRASTRUCTURE ONLY.

Purpose: the reference cannot be built or imported here and its tests pin nothing on this
path, so the C++ restatement in oracle/rolo_oracle.cpp is pinned against this second, independently written
statement of the same algorithm that leans on numpy/scipy library routines instead of hand-rolled ones
(scipy cKDTree, numpy.linalg.svd/inv/solve, scipy Rotation / expm). tests/golden/make_golden.py runs it in the
authoring container and commits its outputs as fixtures; nothing on the GPU box imports this file.

Citations are relative to /root/reference.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation
from scipy.linalg import expm


def knn_exact(xyz32: np.ndarray, k: int = 20):
    """Exact kNN under the oracle's definition: float32 d2 = ((dx*dx)+(dy*dy))+(dz*dz), order (d2, index)."""
    n = xyz32.shape[0]
    tree = cKDTree(xyz32.astype(np.float64))
    kk = min(n, k + 6)
    _, cand = tree.query(xyz32.astype(np.float64), k=kk)
    p = xyz32[:, None, :]
    q = xyz32[cand]
    d = (p - q).astype(np.float32)
    d2 = ((d[..., 0] * d[..., 0]) + (d[..., 1] * d[..., 1])) + (d[..., 2] * d[..., 2])
    order = np.lexsort((cand, d2), axis=1)
    idx = np.take_along_axis(cand, order, axis=1)[:, :k]
    dd = np.take_along_axis(d2, order, axis=1)
    # the (k+1)-th candidate must be strictly worse than what a wider search could add
    assert kk == n or np.all(dd[:, k - 1] <= dd[:, -1])
    return idx.astype(np.int32), dd[:, :k]


def covariances(xyz32: np.ndarray, k: int = 20, values=(1.0, 1.0, 1e-3)):
    """rot_vgicp_impl.hpp:421-496 with RegularizationMethod::PLANE."""
    idx, _ = knn_exact(xyz32, k)
    nb = xyz32[idx].astype(np.float64)            # n,k,3
    c = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c) / k
    U, _, Vt = np.linalg.svd(cov)
    return np.einsum("nik,k,nkj->nij", U, np.asarray(values), Vt), idx


def polar_keys(x: np.ndarray, res):
    """vmp_voxel.hpp:208-211."""
    r = np.sqrt((x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]) + x[:, 2] * x[:, 2])
    k0 = np.floor((np.arctan2(x[:, 1], x[:, 0]) + np.pi) / res[0])
    k1 = np.floor(np.arccos(x[:, 2] / r) / res[1])
    k2 = np.floor(r / res[2])
    return np.stack([k0, k1, k2], axis=1).astype(np.int32)


def uniform_keys(x: np.ndarray, leaf: float):
    """vmp_voxel.hpp:199-201."""
    return np.floor(x / leaf - 0.5).astype(np.int32)


class Twin:
    def __init__(self, source4: np.ndarray, target4: np.ndarray, voxel_type="polar", polar_res=(0.175, 0.175, 2.0),
                 leaf=1.0, k=20, max_iterations=64, rot_eps=2e-3, trans_eps=5e-4, lm_max=10, lm_init=1e-9,
                 fixed_iterations=0, q2_intended=False):
        self.src = source4[:, :3].astype(np.float32)
        self.tgt = target4[:, :3].astype(np.float32)
        self.voxel_type, self.polar_res, self.leaf, self.k = voxel_type, np.asarray(polar_res, float), leaf, k
        self.max_iterations, self.rot_eps, self.trans_eps = max_iterations, rot_eps, trans_eps
        self.lm_max, self.lm_init, self.fixed_iterations, self.q2_intended = lm_max, lm_init, fixed_iterations, q2_intended
        self.src_cov, self.src_knn = covariances(self.src, k)
        self.tgt_cov, self.tgt_knn = covariances(self.tgt, k)
        self._build_map()
        self.trace = []

    def keys(self, x):
        return polar_keys(x, self.polar_res) if self.voxel_type == "polar" else uniform_keys(x, self.leaf)

    def _build_map(self):
        """vmp_voxel.hpp:167-197; voxel ids in order of first appearance."""
        x = self.tgt.astype(np.float64)
        keys = self.keys(x)
        self.tgt_keys = keys
        uniq, first, inv = np.unique(keys, axis=0, return_index=True, return_inverse=True)
        inv = inv.reshape(-1)
        order = np.argsort(first, kind="stable")
        rank = np.empty_like(order); rank[order] = np.arange(order.size)
        vid = rank[inv]
        V = uniq.shape[0]
        self.vox_keys = uniq[order]
        self.vox_count = np.bincount(vid, minlength=V).astype(np.int32)
        mean = np.zeros((V, 3)); cov = np.zeros((V, 3, 3))
        np.add.at(mean, vid, x)
        np.add.at(cov, vid, self.tgt_cov)
        self.vox_mean = mean / self.vox_count[:, None]
        self.vox_cov = cov / self.vox_count[:, None, None]
        self.lookup = {tuple(k): i for i, k in enumerate(self.vox_keys.tolist())}

    def correspond(self, T):
        """rot_vgicp_impl.hpp:173-222 with DIRECT1."""
        R, t = T[:3, :3], T[:3, 3]
        p = self.src.astype(np.float64)
        tp = p @ R.T + t
        keys = self.keys(tp)
        vid = np.array([self.lookup.get(tuple(k), -1) for k in keys.tolist()], np.int64)
        sel = np.nonzero(vid >= 0)[0]
        self.c_src, self.c_vox = sel, vid[sel]
        RCR = self.vox_cov[self.c_vox] + R @ self.src_cov[sel] @ R.T
        self.M = np.linalg.inv(RCR)

    def _residuals(self, T):
        R, t = T[:3, :3], T[:3, 3]
        tp = self.src[self.c_src].astype(np.float64) @ R.T + t
        e = self.vox_mean[self.c_vox] - tp
        w = np.sqrt(self.vox_count[self.c_vox].astype(np.float64))
        return tp, e, w

    @staticmethod
    def _skew(v):
        S = np.zeros(v.shape[:-1] + (3, 3))
        S[..., 0, 1] = -v[..., 2]; S[..., 0, 2] = v[..., 1]
        S[..., 1, 0] = v[..., 2];  S[..., 1, 2] = -v[..., 0]
        S[..., 2, 0] = -v[..., 1]; S[..., 2, 1] = v[..., 0]
        return S

    def so3_linearize(self, T):
        """rot_vgicp_impl.hpp:293-388."""
        self.correspond(T)
        tp, e, w = self._residuals(T)
        err = float(np.sum(w * np.einsum("ni,nij,nj->n", e, self.M, e)))
        J = self._skew(tp)
        H = np.einsum("n,nki,nkl,nlj->ij", w, J, self.M, J)
        b = np.einsum("n,nki,nkl,nl->i", w, J, self.M, e)
        return err, H, b

    def linearize6(self, T):
        """rot_vgicp_impl.hpp:225-290."""
        self.correspond(T)
        tp, e, w = self._residuals(T)
        err = float(np.sum(w * np.einsum("ni,nij,nj->n", e, self.M, e)))
        J = np.concatenate([self._skew(tp), -np.broadcast_to(np.eye(3), (tp.shape[0], 3, 3))], axis=2)
        H = np.einsum("n,nki,nkl,nlj->ij", w, J, self.M, J)
        b = np.einsum("n,nki,nkl,nl->i", w, J, self.M, e)
        return err, H, b

    def compute_error(self, T):
        """rot_vgicp_impl.hpp:391-417 (cached correspondences / Mahalanobis)."""
        _, e, w = self._residuals(T)
        return float(np.sum(w * np.einsum("ni,nij,nj->n", e, self.M, e)))

    def t3(self, t, g, l, dtn, dtn1, lam32, error_variant, want_H=True):
        """rot_vgicp_impl.hpp:499-607 / :610-658 incl. the SURVEY Q2 restatement."""
        p = self.src[self.c_src].astype(np.float64)
        tp = p + t
        ba = p - g
        e = self.vox_mean[self.c_vox] - tp
        if self.q2_intended:
            last = np.asarray(l, float)
        else:
            last = np.array([1.0, 0, 0]) if error_variant else np.zeros(3)
        ct = (ba - tp) / dtn - last / dtn1
        w = np.sqrt(self.vox_count[self.c_vox].astype(np.float64))
        lam_n = float(np.float32(lam32) / np.float32(len(self.c_src)))
        err = float(np.sum(w * (np.einsum("ni,nij,nj->n", e, self.M, e) + lam_n * np.einsum("ni,nij,nj->n", ct, self.M, ct))))
        if not want_H:
            return err
        J1 = np.concatenate([self._skew(tp), -np.broadcast_to(np.eye(3), (tp.shape[0], 3, 3))], axis=2)
        J2 = J1 / dtn
        H = np.einsum("n,nki,nkl,nlj->ij", w, J1, self.M, J1) + lam_n * np.einsum("n,nki,nkl,nlj->ij", w, J2, self.M, J2)
        b = np.einsum("n,nki,nkl,nl->i", w, J1, self.M, e) + lam_n * np.einsum("n,nki,nkl,nl->i", w, J2, self.M, ct)
        return err, H, b

    # ---- drivers: lsq_registration_impl.hpp ----
    def align(self, guess=None):
        """:152-179 with rot_step_lm :273-324 (SO3_LevenbergMarquardt)."""
        x0 = np.eye(4) if guess is None else np.asarray(guess, np.float32).astype(np.float64)
        lam = -1.0
        conv = False
        it = 0
        maxit = self.fixed_iterations if self.fixed_iterations > 0 else self.max_iterations
        hist = []
        for i in range(maxit):
            if not self.fixed_iterations and conv:
                break
            it = i + 1
            y0, H, b = self.so3_linearize(x0)
            hist.append((y0, H.copy(), b.copy()))
            if lam < 0:
                lam = self.lm_init * np.abs(np.diag(H)).max()
            nu = 2.0
            ok = False
            delta = np.eye(4)
            for j in range(self.lm_max):
                d = np.linalg.solve(H + lam * np.eye(3), -b)
                delta = np.eye(4)
                delta[:3, :3] = Rotation.from_rotvec(d).as_matrix()
                xi = delta @ x0
                yi = self.compute_error(xi)
                rho = (y0 - yi) / (d @ (lam * d - b))
                rconv = np.abs(delta[:3, :3] - np.eye(3)).max() / self.rot_eps < 1
                if rho < 0:
                    if rconv:
                        self.trace.append((0, i, j, 2, y0, yi, rho, lam)); ok = True; break
                    self.trace.append((0, i, j, 0, y0, yi, rho, lam))
                    lam *= nu; nu *= 2
                    continue
                self.trace.append((0, i, j, 1, y0, yi, rho, lam))
                x0 = xi
                lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                ok = True
                break
            if not ok:
                break
            conv = np.abs(delta[:3, :3] - np.eye(3)).max() / self.rot_eps < 1
        return x0, it, conv, hist

    def compute_translation(self, trans, g, l, dtn=0.1, dtn1=0.1, ct_lambda=0.3):
        """:55-80 with step_t_optimize :84-139."""
        t0 = np.array(trans, float); g = np.asarray(g, float); l = np.asarray(l, float)
        lam = -1.0; conv = False; it = 0
        hist = []
        for i in range(self.max_iterations):
            if conv:
                break
            it = i + 1
            y0, H, b = self.t3(t0, g, l, dtn, dtn1, ct_lambda, False)
            hist.append((y0, H.copy(), b.copy()))
            if lam < 0:
                lam = self.lm_init * np.abs(np.diag(H)).max()
            nu = 2.0; ok = False; delta = np.zeros(3)
            for j in range(self.lm_max):
                d = np.linalg.solve(H + lam * np.eye(6), -b)
                xi_mat = np.zeros((4, 4)); xi_mat[:3, :3] = self._skew(d[:3]); xi_mat[:3, 3] = d[3:]
                delta = expm(xi_mat)[:3, 3]  # se3_exp(d).translation()
                xi = delta + t0
                yi = self.t3(xi, g, l, dtn, dtn1, ct_lambda, True, want_H=False)
                rho = (y0 - yi) / (d @ (lam * d - b))
                tconv = np.abs(delta).max() / self.trans_eps < 1
                if rho < 0:
                    if tconv:
                        self.trace.append((1, i, j, 2, y0, yi, rho, lam)); ok = True; break
                    self.trace.append((1, i, j, 0, y0, yi, rho, lam))
                    lam *= nu; nu *= 2
                    continue
                self.trace.append((1, i, j, 1, y0, yi, rho, lam))
                t0 = xi
                lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                ok = True
                break
            if not ok:
                break
            conv = np.abs(delta).max() / self.trans_eps < 1
        return t0, it, hist
