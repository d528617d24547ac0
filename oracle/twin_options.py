"""ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED by the reference itself.

The options of RotVGICP that the reference never selects but exposes — the five other RegularizationMethods
(rot_vgicp_impl.hpp:458-488), DIRECT7 / DIRECT27 neighbour offsets (vmp_voxel.hpp:13-47, update_correspondences :173-222)
— restated independently in numpy on top of oracle/twin.py, so that the C++ oracle's versions have a second opinion too.
"""
from __future__ import annotations

import numpy as np

from .twin import Twin, knn_exact

NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS, PLANE_S = range(6)   # gicp_settings.hpp:6-13


def covariances(xyz32, method, k=20):
    """calculate_covariances :421-496 for any RegularizationMethod."""
    idx, _ = knn_exact(xyz32, k)
    nb = xyz32[idx].astype(np.float64)
    c = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c) / k
    if method == NONE:
        return cov
    if method == FROBENIUS:
        Ci = np.linalg.inv(cov + 1e-3 * np.eye(3))
        nrm = np.sqrt((Ci * Ci).sum(axis=(1, 2)))                           # Matrix::norm() = Frobenius
        return np.linalg.inv(Ci / nrm[:, None, None])
    U, s, Vt = np.linalg.svd(cov)
    if method == PLANE:
        v = np.tile(np.array([1.0, 1.0, 1e-3]), (s.shape[0], 1))
    elif method == MIN_EIG:
        v = np.maximum(s, 1e-3)
    elif method == NORMALIZED_MIN_EIG:
        v = np.maximum(s / s.max(axis=1, keepdims=True), 1e-3)
    elif method == PLANE_S:
        v = s / s.sum(axis=1, keepdims=True); v[:, 2] = 1e-3
    else:
        raise ValueError(method)
    return np.einsum("nik,nk,nkj->nij", U, v, Vt)


def neighbor_offsets(n_off):
    if n_off == 1:
        return np.array([[0, 0, 0]])
    if n_off == 7:
        return np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    return np.array([[i - 1, j - 1, k - 1] for i in range(3) for j in range(3) for k in range(3)])


def so3_linearize_multi(tw: Twin, T, n_off):
    """update_correspondences with DIRECT7 / DIRECT27 (every occupied neighbouring voxel is a correspondence of its own, in
    source order then offset order) followed by so3_linearize :293-388. Returns err, H, b and the (source, voxel key) list."""
    R, t = T[:3, :3], T[:3, 3]
    tp = tw.src.astype(np.float64) @ R.T + t
    keys = tw.keys(tp)
    src_i, vox = [], []
    for i, k in enumerate(keys.tolist()):
        for o in neighbor_offsets(n_off).tolist():
            v = tw.lookup.get((k[0] + o[0], k[1] + o[1], k[2] + o[2]), -1)
            if v >= 0:
                src_i.append(i); vox.append(v)
    src_i = np.array(src_i, np.int64); vox = np.array(vox, np.int64)
    M = np.linalg.inv(tw.vox_cov[vox] + R @ tw.src_cov[src_i] @ R.T)
    e = tw.vox_mean[vox] - tp[src_i]
    w = np.sqrt(tw.vox_count[vox].astype(np.float64))
    err = float(np.sum(w * np.einsum("ni,nij,nj->n", e, M, e)))
    J = Twin._skew(tp[src_i])
    H = np.einsum("n,nki,nkl,nlj->ij", w, J, M, J)
    b = np.einsum("n,nki,nkl,nl->i", w, J, M, e)
    return err, H, b, src_i, tw.vox_keys[vox]


def _se3_exp(d):
    """so3.hpp:80-103 as a matrix exponential (scipy expm of the 4x4 twist)."""
    from scipy.linalg import expm
    X = np.zeros((4, 4)); X[:3, :3] = Twin._skew(np.asarray(d[:3], float)); X[:3, 3] = d[3:]
    return expm(X)


def align6(tw: Twin, optimizer, guess=None):
    """computeTransformation :152-179 with step_lm :225-270 ("lm") or step_gn :208-222 ("gn"); is_converged :182-191.
    Returns the final transformation (float-rounded like final_transformation_), the outer iterations and the converged flag."""
    x0 = np.eye(4) if guess is None else np.asarray(guess, np.float32).astype(np.float64)
    lam = -1.0
    conv = False; it = 0

    def converged(delta):
        r = np.abs(delta[:3, :3] - np.eye(3)).max() / tw.rot_eps
        t = np.abs(delta[:3, 3]).max() / tw.trans_eps
        return max(r, t) < 1

    for i in range(tw.max_iterations):
        if conv:
            break
        it = i + 1
        y0, H, b = tw.linearize6(x0)
        if optimizer == "gn":
            delta = _se3_exp(np.linalg.solve(H, -b))
            x0 = delta @ x0
        else:
            if lam < 0:
                lam = tw.lm_init * np.abs(np.diag(H)).max()
            nu = 2.0; ok = False
            for _ in range(tw.lm_max):
                d = np.linalg.solve(H + lam * np.eye(6), -b)
                delta = _se3_exp(d)
                xi = delta @ x0
                yi = tw.compute_error(xi)
                rho = (y0 - yi) / (d @ (lam * d - b))
                if rho < 0:
                    if converged(delta):
                        ok = True; break
                    lam *= nu; nu *= 2
                    continue
                x0 = xi
                lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                ok = True
                break
            if not ok:
                break
        conv = converged(delta)
    return x0, it, conv
