"""TEST INFRASTRUCTURE — independent numpy/scipy statement of the back end's scan-to-submap optimisation (reference
src/backMapping.cpp: scan2MapOptimization :681-711, cornerOptimization :720-824, surfOptimization :827-901, combineOptimizationCoeffs
:904-925, LMOptimization :929-1058), written from the reference source on library routines: scipy cKDTree for the 5-NN search
(pcl::KdTreeFLANN::nearestKSearch), numpy.linalg.eigh in float32 for cv::eigen, numpy.linalg.lstsq for colPivHouseholderQr().solve,
numpy.linalg.qr / solve in float32 for cv::solve(DECOMP_QR). Float32 where the reference is float.
"""
import numpy as np
from scipy.spatial import cKDTree

f32 = np.float32


def get_transformation(x, y, z, roll, pitch, yaw):   # pcl::getTransformation, float
    A, B, C, D, E, F = (f32(v) for v in (np.cos(f32(yaw)), np.sin(f32(yaw)), np.cos(f32(pitch)), np.sin(f32(pitch)), np.cos(f32(roll)), np.sin(f32(roll))))
    T = np.eye(4, dtype=f32)
    T[0, :3] = [A * C, A * D * F - B * E, B * F + A * D * E]; T[1, :3] = [B * C, A * E + B * D * F, B * D * E - A * F]; T[2, :3] = [-D, C * F, C * E]
    T[:3, 3] = [x, y, z]
    return T


def associate(T, pts):   # pointAssociateToMap :293-299, float, left to right
    p = pts[:, :3].astype(f32)
    return np.stack([T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1] + T[r, 2] * p[:, 2] + T[r, 3] for r in range(3)], axis=1).astype(f32)


def knn5(tree_pts, tree, q):
    """exact 5-NN with float squared distances ((dx*dx)+(dy*dy))+(dz*dz), ties by index"""
    _, idx = tree.query(q.astype(np.float64), k=min(12, tree_pts.shape[0]))
    d = q[:, None, :] - tree_pts[idx]                               # float32
    d2 = ((d[..., 0] * d[..., 0]) + (d[..., 1] * d[..., 1])) + (d[..., 2] * d[..., 2])
    order = np.lexsort((idx, d2), axis=1)[:, :5]
    return np.take_along_axis(idx, order, 1), np.take_along_axis(d2, order, 1)


def corner_coeffs(sel_pts, mp, idx, d2):
    n = sel_pts.shape[0]
    coeff = np.zeros((n, 4), f32); flag = np.zeros(n, bool)
    ok = d2[:, 4] < f32(1.0)
    P = mp[idx]                                                     # n x 5 x 3
    c = np.zeros((n, 3), f32)
    for j in range(5):
        c += P[:, j]
    c /= f32(5)
    a = P - c[:, None, :]
    cov = np.zeros((n, 3, 3), f32)
    for j in range(5):
        cov += a[:, j, :, None] * a[:, j, None, :]
    cov /= f32(5)
    w, v = np.linalg.eigh(cov)                                      # float32, ascending
    d0, d1 = w[:, 2], w[:, 1]
    dirv = v[:, :, 2]                                               # eigenvector of the largest eigenvalue (sign is irrelevant below)
    line = ok & (d0 > f32(3) * d1)
    x0, y0, z0 = sel_pts[:, 0], sel_pts[:, 1], sel_pts[:, 2]
    x1, y1, z1 = (c[:, k] + f32(0.1) * dirv[:, k] for k in range(3))
    x2, y2, z2 = (c[:, k] - f32(0.1) * dirv[:, k] for k in range(3))
    m1 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1); m2 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1); m3 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)
    with np.errstate(all="ignore"):
        a012 = np.sqrt(m1 * m1 + m2 * m2 + m3 * m3); l12 = np.sqrt((x1 - x2) ** 2 + (y1 - y2) ** 2 + (z1 - z2) ** 2)
        la = ((y1 - y2) * m1 + (z1 - z2) * m2) / a012 / l12
        lb = -((x1 - x2) * m1 - (z1 - z2) * m3) / a012 / l12
        lc = -((x1 - x2) * m2 + (y1 - y2) * m3) / a012 / l12
        ld2 = a012 / l12
        s = f32(1) - f32(0.9) * np.abs(ld2)
    flag = line & (s > f32(0.1))
    coeff[:, 0] = s * la; coeff[:, 1] = s * lb; coeff[:, 2] = s * lc; coeff[:, 3] = s * ld2
    coeff[~flag] = 0
    return flag, coeff


def surf_coeffs(ori, sel_pts, mp, idx, d2):
    n = sel_pts.shape[0]
    coeff = np.zeros((n, 4), f32)
    ok = d2[:, 4] < f32(1.0)
    P = mp[idx].astype(np.float64)
    # least squares of the 5 x 3 systems A x = -1 (batched normal equations in float64, narrowed to float: the systems are well conditioned)
    AtA = np.einsum("nji,njk->nik", P, P); Atb = -P.sum(axis=1)
    with np.errstate(all="ignore"):
        try:
            x = np.linalg.solve(AtA, Atb[..., None])[..., 0]
        except np.linalg.LinAlgError:
            x = np.stack([np.linalg.lstsq(P[i], -np.ones(5), rcond=None)[0] for i in range(n)])
        x = x.astype(f32)
        ps = np.sqrt(x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1] + x[:, 2] * x[:, 2])
        pa, pb, pc, pd = x[:, 0] / ps, x[:, 1] / ps, x[:, 2] / ps, f32(1) / ps
        Pf = mp[idx]
        dist = np.abs(pa[:, None] * Pf[:, :, 0] + pb[:, None] * Pf[:, :, 1] + pc[:, None] * Pf[:, :, 2] + pd[:, None])
        valid = ok & np.all(dist <= f32(0.2), axis=1) & np.isfinite(ps)
        pd2 = pa * sel_pts[:, 0] + pb * sel_pts[:, 1] + pc * sel_pts[:, 2] + pd
        o = ori[:, :3].astype(f32)
        s = f32(1) - f32(0.9) * np.abs(pd2) / np.sqrt(np.sqrt(o[:, 0] * o[:, 0] + o[:, 1] * o[:, 1] + o[:, 2] * o[:, 2]))
    flag = valid & (s > f32(0.1))
    coeff[:, 0] = s * pa; coeff[:, 1] = s * pb; coeff[:, 2] = s * pc; coeff[:, 3] = s * pd2
    coeff[~flag] = 0
    return flag, coeff


def scan2map(corner, surf, map_corner, map_surf, tf, edge_min=10, surf_min=100):
    """Returns (transformTobeMapped, stats dict, selected flags, coeffs of the last iteration)."""
    tf = np.asarray(tf, f32).copy()
    corner = np.asarray(corner, f32); surf = np.asarray(surf, f32)
    mc = np.asarray(map_corner, f32)[:, :3]; ms = np.asarray(map_surf, f32)[:, :3]
    st = dict(skipped=0, iterations=0, converged=0, degenerate=0, n_selected=0)
    if not (corner.shape[0] > edge_min and surf.shape[0] > surf_min):
        st["skipped"] = 1
        return tf, st, None, None
    tc, ts = cKDTree(mc.astype(np.float64)), cKDTree(ms.astype(np.float64))
    degenerate = False; matP = np.eye(6, dtype=f32); flags = coeffs = None
    for it in range(30):
        T = get_transformation(tf[3], tf[4], tf[5], tf[0], tf[1], tf[2])
        sc, ss = associate(T, corner), associate(T, surf)
        ic, dc = knn5(mc, tc, sc); isf, ds = knn5(ms, ts, ss)
        fc, cc = corner_coeffs(sc, mc, ic, dc); fs, cs = surf_coeffs(surf, ss, ms, isf, ds)
        flags = np.concatenate([fc, fs]); coeffs = np.concatenate([cc, cs])
        ori = np.concatenate([corner[fc, :3], surf[fs, :3]]).astype(f32); co = np.concatenate([cc[fc], cs[fs]])
        st["iterations"] = it + 1; st["n_selected"] = int(ori.shape[0])
        if ori.shape[0] < 50:
            break
        srx, crx, sry, cry, srz, crz = (f32(v) for v in (np.sin(tf[1]), np.cos(tf[1]), np.sin(tf[2]), np.cos(tf[2]), np.sin(tf[0]), np.cos(tf[0])))
        ox, oy, oz = ori[:, 1], ori[:, 2], ori[:, 0]; kx, ky, kz = co[:, 1], co[:, 2], co[:, 0]
        arx = (crx * sry * srz * ox + crx * crz * sry * oy - srx * sry * oz) * kx + (-srx * srz * ox - crz * srx * oy - crx * oz) * ky + (crx * cry * srz * ox + crx * cry * crz * oy - cry * srx * oz) * kz
        ary = ((cry * srx * srz - crz * sry) * ox + (sry * srz + cry * crz * srx) * oy + crx * cry * oz) * kx + ((-cry * crz - srx * sry * srz) * ox + (cry * srz - crz * srx * sry) * oy - crx * sry * oz) * kz
        arz = ((crz * srx * sry - cry * srz) * ox + (-cry * crz - srx * sry * srz) * oy) * kx + (crx * crz * ox - crx * srz * oy) * ky + ((sry * srz + cry * crz * srx) * ox + (crz * sry - cry * srx * srz) * oy) * kz
        A = np.stack([arz, arx, ary, kz, kx, ky], axis=1).astype(np.float64); b = -co[:, 3].astype(np.float64)
        AtA = (A.T @ A).astype(f32); AtB = (A.T @ b).astype(f32)
        try:
            X = np.linalg.solve(AtA.astype(np.float64), AtB.astype(np.float64)).astype(f32)
        except np.linalg.LinAlgError:
            X = np.zeros(6, f32)
        if it == 0:
            E, V = np.linalg.eigh(AtA.astype(np.float64))
            E = E[::-1].astype(f32); V = V[:, ::-1].T.astype(f32)   # descending, eigenvectors as rows
            V2 = V.copy(); degenerate = False
            for i in range(5, -1, -1):
                if E[i] < 100:
                    V2[i] = 0; degenerate = True
                else:
                    break
            matP = (np.linalg.inv(V.astype(np.float64)) @ V2.astype(np.float64)).astype(f32)
        if degenerate:
            X = (matP @ X).astype(f32)
        tf = (tf + X).astype(f32)
        dR = np.sqrt(np.sum(np.rad2deg(X[:3].astype(np.float64)) ** 2)); dT = np.sqrt(np.sum((X[3:].astype(np.float64) * 100) ** 2))
        if dR < 0.05 and dT < 0.05:
            st["converged"] = 1
            break
    st["degenerate"] = int(degenerate)
    return tf, st, flags, coeffs
