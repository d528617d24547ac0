"""ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED by the reference itself (it ships no vectors for this path).

Independent numpy statement of the front half of the hot path, written from the reference sources and from the documented
behaviour of PCL — NOT from oracle/rolo_oracle_front.cpp — so that the C++ oracle has a second opinion:

  project()           ImageProjection::projectPointCloud + cloudExtraction   (src/imageProjection.cpp:399-505)
  extract_features()  FeatureExtraction::calculateSmoothness / markOccludedPoints / extractFeatures
                                                                             (src/featureExtraction.cpp:87-266)
  voxel_grid()        pcl::VoxelGrid<PointXYZI>::filter as used at :254-258
  pose algebra        pcl::getTransformation / getTranslationAndEulerAngles, the float pose chain of
                      LidarOdometry::updateTransform (src/lidarOdometry.cpp:572-626)

Where the reference loops serially this file vectorises with numpy where the result cannot depend on the order (first
occurrence per pixel, stencils) and keeps plain Python loops where it can (the greedy picks). float32 arithmetic follows the
reference's expressions operation by operation (numpy does not contract to FMA).
Conventions shared with the C++ oracle because the reference leaves them open: per-frame arrays start at zero,
std::sort ties are broken by index (Q7).
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32


def c_round(v):
    """C round(): halves away from zero (numpy rounds halves to even)."""
    return np.sign(v) * np.floor(np.abs(v) + 0.5)


def project(xyz, ring, n_scan, horizon_scan, downsample_rate=1, min_range=2.0, max_range=1000.0):
    xyz = np.asarray(xyz, F); ring = np.asarray(ring).astype(np.int64)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rng = np.sqrt(x * x + y * y + z * z)                                   # utility.h:462-465, float
    keep = ~((rng < F(min_range)) | (rng > F(max_range)))                   # :413 (NaN passes here ...)
    keep &= (ring >= 0) & (ring < n_scan) & (ring % downsample_rate == 0)   # :426-431
    ang = (np.arctan2(x, y) * F(180)).astype(np.float64) / math.pi          # :437: float atan2, float * 180, / M_PI in double
    ang = ang.astype(F)                                                     # stored in a float
    ang_res_x = F(360.0 / float(F(horizon_scan)))                            # :438 static float
    with np.errstate(invalid="ignore"):
        col = -c_round((ang.astype(np.float64) - 90.0) / np.float64(ang_res_x)) + horizon_scan // 2   # :440, double
    col = np.where(np.isnan(col), -1e18, col)                               # ... and dies here: int(NaN) is INT_MIN on x86
    col = np.trunc(col).astype(np.int64)
    col = np.where(col >= horizon_scan, col - horizon_scan, col)            # :443-444
    keep &= (col >= 0) & (col < horizon_scan)
    idx = np.nonzero(keep)[0]
    pix = ring[idx] * horizon_scan + col[idx]
    # "first point to claim a pixel wins" (:451): the first occurrence in firing order
    upix, first = np.unique(pix, return_index=True)
    owner = idx[first]                                                      # raw index per filled pixel, ascending pixel = row-major
    row = upix // horizon_scan
    extracted = np.zeros((owner.size, 4), F)
    extracted[:, :3] = xyz[owner]
    extracted[:, 3] = ring[owner].astype(F) * z[owner]                      # :410 intensity <- ring * z
    counts = np.bincount(row, minlength=n_scan)
    ends = np.cumsum(counts)
    start_ring = (ends - counts) - 1 + 5                                    # :485
    end_ring = ends - 1 - 5                                                 # :503
    return dict(n=int(owner.size), extracted=extracted, point_col_ind=(upix % horizon_scan).astype(np.int32), point_range=rng[owner].astype(F),
                start_ring=start_ring.astype(np.int32), end_ring=end_ring.astype(np.int32), owner=owner)


def voxel_grid(pts, leaf):
    """pcl::VoxelGrid (downsample_all_data, no limit filter): cell = floor(p / leaf) - min cell, linear index, points sorted by
    (index, input order), centroid of every run accumulated in that order in float (pcl::CentroidPoint), one output per run."""
    pts = np.asarray(pts, F)
    if pts.shape[0] == 0:
        return pts.reshape(0, 4)
    inv = F(1.0) / F(leaf)
    mn = pts[:, :3].min(0); mx = pts[:, :3].max(0)
    span = ((mx - mn) * inv).astype(np.int64) + 1
    if int(span[0]) * int(span[1]) * int(span[2]) > np.iinfo(np.int32).max:
        return pts.copy()                                                   # PCL warns and passes the cloud through
    min_b = np.floor(mn * inv).astype(np.int64); max_b = np.floor(mx * inv).astype(np.int64)
    div = max_b - min_b + 1
    ijk = np.floor(pts[:, :3] * inv).astype(np.int64) - min_b
    cell = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.lexsort((np.arange(pts.shape[0]), cell))
    out = []
    i = 0
    while i < order.size:
        j = i
        acc = np.zeros(4, F)
        while j < order.size and cell[order[j]] == cell[order[i]]:
            acc = acc + pts[order[j]]                                       # float accumulators, sorted order
            j += 1
        out.append(acc / F(j - i))
        i = j
    return np.array(out, F)


def extract_features(proj, n_scan, edge_threshold=0.8, surf_threshold=0.1, leaf=0.4):
    r = proj["point_range"]; col = proj["point_col_ind"].astype(np.int64); pts = proj["extracted"]
    n = proj["n"]
    G = 8                                                                   # guard cells: the reference indexes a few cells past both ends
    curv = np.zeros(n + 2 * G, F); picked = np.zeros(n + 2 * G, np.int32); label = np.zeros(n + 2 * G, np.int32)
    colg = np.zeros(n + 2 * G, np.int64); colg[G:G + n] = col
    sm_val = np.zeros(n + 2 * G, F); sm_ind = np.zeros(n + 2 * G, np.int64)   # cloudSmoothness, zero-initialised (Q6)
    if n > 10:
        i = np.arange(5, n - 5)
        d = r[i - 5] + r[i - 4]
        d = d + r[i - 3]; d = d + r[i - 2]; d = d + r[i - 1]
        d = d - r[i] * F(10)
        d = d + r[i + 1]; d = d + r[i + 2]; d = d + r[i + 3]; d = d + r[i + 4]; d = d + r[i + 5]   # :93-98, written order
        curv[G + i] = d * d
        sm_val[G + i] = curv[G + i]; sm_ind[G + i] = i
    # markOccludedPoints :112-150 — every mark stores 1, so the loop order is irrelevant
    for i in range(5, n - 6):
        d1, d2 = r[i], r[i + 1]
        if abs(int(col[i + 1] - col[i])) < 10:
            if float(d1 - d2) > 0.3:
                picked[G + i - 5:G + i + 1] = 1
            elif float(d2 - d1) > 0.3:
                picked[G + i + 1:G + i + 7] = 1
        diff1 = abs(F(r[i - 1] - r[i])); diff2 = abs(F(r[i + 1] - r[i]))
        if float(diff1) > 0.02 * float(r[i]) and float(diff2) > 0.02 * float(r[i]):
            picked[G + i] = 1

    def mark(ind):
        picked[G + ind] = 1
        for l in range(1, 6):
            if abs(int(colg[G + ind + l] - colg[G + ind + l - 1])) > 10:
                break
            picked[G + ind + l] = 1
        for l in range(-1, -6, -1):
            if abs(int(colg[G + ind + l] - colg[G + ind + l + 1])) > 10:
                break
            picked[G + ind + l] = 1

    corners, surfaces = [], []
    for ring in range(n_scan):
        s, e = int(proj["start_ring"][ring]), int(proj["end_ring"][ring])
        scan = []
        for j in range(6):
            sp = (s * (6 - j) + e * j) // 6 if (s * (6 - j) + e * j) >= 0 else -((-(s * (6 - j) + e * j)) // 6)   # C division truncates
            t2 = s * (5 - j) + e * (j + 1)
            ep = (t2 // 6 if t2 >= 0 else -((-t2) // 6)) - 1
            if sp >= ep:
                continue
            # std::sort(begin + sp, begin + ep): ep is NOT part of the sorted range but is part of both loops (Q7)
            seg = sorted(range(sp, ep), key=lambda k: (float(sm_val[G + k]), int(sm_ind[G + k])))
            sm_val[G + sp:G + ep], sm_ind[G + sp:G + ep] = sm_val[G + np.array(seg)].copy(), sm_ind[G + np.array(seg)].copy()
            picked_n = 0
            for k in range(ep, sp - 1, -1):
                ind = int(sm_ind[G + k])
                if picked[G + ind] == 0 and curv[G + ind] > F(edge_threshold):
                    picked_n += 1
                    if picked_n <= 20:
                        label[G + ind] = 1
                        corners.append(pts[ind])
                    else:
                        break
                    mark(ind)
            for k in range(sp, ep + 1):
                ind = int(sm_ind[G + k])
                if picked[G + ind] == 0 and curv[G + ind] < F(surf_threshold):
                    label[G + ind] = -1
                    mark(ind)
            for k in range(sp, ep + 1):
                if label[G + k] <= 0:
                    scan.append(pts[k])
        if scan:
            surfaces.append(voxel_grid(np.array(scan, F), leaf))
    corner = np.array(corners, F).reshape(-1, 4)
    surface = np.concatenate(surfaces).astype(F) if surfaces else np.zeros((0, 4), F)
    return dict(corner=corner, surface=surface, curvature=curv[G:G + n].copy(), picked=picked[G:G + n].copy(), label=label[G:G + n].copy())


def azimuth_times(xyz, scan_period=0.1):
    """deskewCloudInfo without a time field (imageProjection.cpp:270-327), vectorised: halfPassed flips at the first point whose
    (first-half-adjusted) azimuth is more than pi past the start, so the serial flag is a prefix property."""
    xyz = np.asarray(xyz, F)
    two_pi = 2 * math.pi
    start = F(-np.arctan2(xyz[0, 1], xyz[0, 0]))
    end = F(np.float64(F(-np.arctan2(xyz[-1, 1], xyz[-1, 0]))) + two_pi)
    if np.float64(F(end - start)) > 3 * math.pi:
        end = F(np.float64(end) - two_pi)
    elif np.float64(F(end - start)) < math.pi:
        end = F(np.float64(end) + two_pi)
    diff = F(end - start)
    ori = (-np.arctan2(xyz[:, 1], xyz[:, 0])).astype(F)
    o64 = ori.astype(np.float64)
    a = np.where(o64 < np.float64(start) - math.pi / 2, (o64 + two_pi).astype(F), np.where(o64 > np.float64(start) + math.pi * 3 / 2, (o64 - two_pi).astype(F), ori)).astype(F)
    passed = (a - start).astype(F).astype(np.float64) > math.pi
    k = int(np.argmax(passed)) if passed.any() else xyz.shape[0]            # the point that sets halfPassed still uses the first rule
    b = (o64 + two_pi).astype(F); b64 = b.astype(np.float64)
    b = np.where(b64 < np.float64(end) - math.pi * 3 / 2, (b64 + two_pi).astype(F), np.where(b64 > np.float64(end) + math.pi / 2, (b64 - two_pi).astype(F), b)).astype(F)
    final = np.where(np.arange(xyz.shape[0]) <= k, a, b).astype(F)
    return (F(scan_period) * ((final - start).astype(F) / diff).astype(F)).astype(F)


# ---- float pose algebra of the odometry node ---------------------------------------------------------------------------
def get_transformation(x, y, z, roll, pitch, yaw):
    """pcl::getTransformation (float): R = Rz(yaw) Ry(pitch) Rx(roll)."""
    A, B = F(np.cos(F(yaw))), F(np.sin(F(yaw))); Cc, D = F(np.cos(F(pitch))), F(np.sin(F(pitch))); E, Fs = F(np.cos(F(roll))), F(np.sin(F(roll)))
    DE, DF = D * E, D * Fs
    return np.array([[A * Cc, A * DF - B * E, B * Fs + A * DE, F(x)],
                     [B * Cc, A * E + B * DF, B * DE - A * Fs, F(y)],
                     [-D, Cc * Fs, Cc * E, F(z)],
                     [0, 0, 0, 1]], F)


def get_translation_and_euler(T):
    """pcl::getTranslationAndEulerAngles (float)."""
    T = np.asarray(T, F)
    return np.array([T[0, 3], T[1, 3], T[2, 3], np.arctan2(T[2, 1], T[2, 2]), np.arcsin(-T[2, 0]), np.arctan2(T[1, 0], T[0, 0])], F)
