"""Deterministic synthetic LiDAR frames for parity tests and bench.py.

Analytic ray-cast of a closed hall: uneven ground z = -1.8 + 0.15 sin(0.2x) cos(0.15y), four walls of a
60 x 40 m box, a ceiling at z = 12 m (so every ray returns and a 128 x 1024 sensor yields exactly 131 072
points), and a fixed table of 12 boxes / cylinders. Gaussian range noise sigma = 0.02 m. Rays are fired at the
azimuth of each range-image column centre, in column-major firing order (ring fastest), and the message
layout is the Velodyne `PointXYZIRT` of reference src/imageProjection.cpp:7-19 (x, y, z, intensity f32,
ring u16, time f32).

This module is data generation only: it contains no registration logic.
"""
from __future__ import annotations

import dataclasses
import numpy as np

SEED = 20260926

SENSORS = {
    # name: (n_scan, horizon_scan, elevation_min_deg, elevation_max_deg)
    "vlp16": (16, 1800, -15.0, 15.0),
    "os1-64": (64, 1024, -22.5, 22.5),
    "os1-128": (128, 1024, -22.5, 22.5),
    "os1-128x2048": (128, 2048, -22.5, 22.5),
    "os1-32x2048": (32, 2048, -22.5, 22.5),     # the geometry of the reference's config/params_os.yaml (N_SCAN 32, Horizon_SCAN 2048)
    "wide-32x4096": (32, 4096, -16.0, 16.0),   # no such sensor: Horizon_SCAN above 2048 (the front end's HBM-scratch path)
}

# fixed obstacle table: ("box", cx, cy, hx, hy, z0, z1) or ("cyl", cx, cy, radius, z0, z1)
OBSTACLES = [
    ("box", 8.0, 5.0, 1.0, 1.5, -2.2, 1.0),
    ("box", -10.0, 7.0, 2.0, 1.0, -2.2, 2.5),
    ("box", 15.0, -9.0, 1.5, 1.5, -2.2, 0.5),
    ("box", -6.0, -11.0, 1.0, 2.5, -2.2, 3.0),
    ("box", 21.0, 12.0, 2.5, 1.0, -2.2, 4.0),
    ("box", -20.0, -6.0, 1.2, 1.2, -2.2, 1.8),
    ("cyl", 5.0, -6.0, 0.6, -2.2, 4.0),
    ("cyl", -4.0, 9.0, 0.4, -2.2, 6.0),
    ("cyl", 12.0, 14.0, 0.8, -2.2, 3.0),
    ("cyl", -15.0, -14.0, 0.5, -2.2, 5.0),
    ("cyl", 24.0, -3.0, 0.7, -2.2, 2.0),
    ("cyl", -24.0, 13.0, 0.9, -2.2, 7.0),
]

WALL_X, WALL_Y, CEIL_Z = 30.0, 20.0, 12.0


def rpy_to_R(roll, pitch, yaw):
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _ground(x, y):
    return -1.8 + 0.15 * np.sin(0.2 * x) * np.cos(0.15 * y)


def _raycast(o, d):
    """o: (3,), d: (M,3) unit directions in world. Returns hit distance (M,), inf for no hit."""
    M = d.shape[0]
    best = np.full(M, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        # walls + ceiling (inside a convex box: exit distance)
        for axis, lim in ((0, WALL_X), (1, WALL_Y)):
            for sgn in (-1.0, 1.0):
                s = (sgn * lim - o[axis]) / d[:, axis]
                p = o[None, :] + s[:, None] * d
                ok = (s > 0) & (np.abs(p[:, 1 - axis]) <= (WALL_Y if axis == 0 else WALL_X) + 1e-9) & (p[:, 2] <= CEIL_Z + 1e-9)
                best = np.where(ok & (s < best), s, best)
        s = (CEIL_Z - o[2]) / d[:, 2]
        p = o[None, :] + s[:, None] * d
        ok = (s > 0) & (np.abs(p[:, 0]) <= WALL_X) & (np.abs(p[:, 1]) <= WALL_Y)
        best = np.where(ok & (s < best), s, best)
        # obstacles
        for ob in OBSTACLES:
            if ob[0] == "box":
                _, cx, cy, hx, hy, z0, z1 = ob
                lo = np.array([cx - hx, cy - hy, z0])
                hi = np.array([cx + hx, cy + hy, z1])
                t1 = (lo[None, :] - o[None, :]) / d
                t2 = (hi[None, :] - o[None, :]) / d
                tmin = np.nanmax(np.minimum(t1, t2), axis=1)
                tmax = np.nanmin(np.maximum(t1, t2), axis=1)
                ok = (tmax >= np.maximum(tmin, 0)) & (tmin > 0)
                best = np.where(ok & (tmin < best), tmin, best)
            else:
                _, cx, cy, rad, z0, z1 = ob
                ox, oy = o[0] - cx, o[1] - cy
                a = d[:, 0] ** 2 + d[:, 1] ** 2
                b = 2 * (ox * d[:, 0] + oy * d[:, 1])
                c = ox * ox + oy * oy - rad * rad
                disc = b * b - 4 * a * c
                sq = np.sqrt(np.where(disc > 0, disc, np.nan))
                s = (-b - sq) / (2 * a)
                z = o[2] + s * d[:, 2]
                ok = (disc > 0) & (s > 0) & (z >= z0) & (z <= z1)
                best = np.where(ok & (s < best), s, best)
                # top cap
                s = (z1 - o[2]) / d[:, 2]
                px, py = o[0] + s * d[:, 0] - cx, o[1] + s * d[:, 1] - cy
                ok = (s > 0) & (px * px + py * py <= rad * rad)
                best = np.where(ok & (s < best), s, best)
    # ground: march then bisect, for downward rays only, up to the current best hit
    down = d[:, 2] < -1e-6
    idx = np.nonzero(down)[0]
    if idx.size:
        dd = d[idx]
        smax = np.minimum(best[idx], 150.0)
        step = 0.5
        nsteps = int(np.ceil(smax.max() / step)) + 1
        s_lo = np.zeros(idx.size)
        found = np.zeros(idx.size, dtype=bool)
        s_hi = np.zeros(idx.size)
        f_prev = o[2] - _ground(o[0], o[1]) + np.zeros(idx.size)
        for k in range(1, nsteps + 1):
            s = np.minimum(k * step, smax)
            f = o[2] + s * dd[:, 2] - _ground(o[0] + s * dd[:, 0], o[1] + s * dd[:, 1])
            newly = (~found) & (f_prev > 0) & (f <= 0)
            s_lo = np.where(newly, np.minimum((k - 1) * step, smax), s_lo)
            s_hi = np.where(newly, s, s_hi)
            found |= newly
            f_prev = np.where(found, f_prev, f)
        for _ in range(48):
            mid = 0.5 * (s_lo + s_hi)
            f = o[2] + mid * dd[:, 2] - _ground(o[0] + mid * dd[:, 0], o[1] + mid * dd[:, 1])
            s_lo = np.where(f > 0, mid, s_lo)
            s_hi = np.where(f > 0, s_hi, mid)
        sg = 0.5 * (s_lo + s_hi)
        upd = found & (sg < best[idx])
        best[idx] = np.where(upd, sg, best[idx])
    return best


@dataclasses.dataclass
class Frame:
    """One raw LiDAR message (Velodyne PointXYZIRT fields) in firing order."""
    xyz: np.ndarray        # (n,3) float32, sensor frame
    intensity: np.ndarray  # (n,) float32
    ring: np.ndarray       # (n,) uint16
    time: np.ndarray       # (n,) float32
    n_scan: int
    horizon_scan: int

    def packed(self):
        """(n, 8) float32 records: x,y,z,intensity,<ring as float>,time,0,0 — 32-byte stride test layout."""
        out = np.zeros((self.xyz.shape[0], 8), np.float32)
        out[:, :3] = self.xyz
        out[:, 3] = self.intensity
        out[:, 4] = self.ring.astype(np.float32)
        out[:, 5] = self.time
        return out


def make_frame(sensor: str, R: np.ndarray, t: np.ndarray, seed: int, noise_sigma: float = 0.02,
               col_stride: int = 1, ring_stride: int = 1) -> Frame:
    """Ray-cast one frame with the sensor at world pose (R, t). `col_stride` / `ring_stride` sub-sample the
    firing pattern (for small test crops); the image geometry (n_scan, horizon_scan) is unchanged."""
    n_scan, H, e0, e1 = SENSORS[sensor]
    rings = np.arange(0, n_scan, ring_stride)
    cols = np.arange(0, H, col_stride)
    elev = np.deg2rad(e0 + (e1 - e0) * rings / (n_scan - 1))
    az = (cols - H // 2) * (2 * np.pi / H)  # column-centre azimuth (imageProjection.cpp:437-442)
    # column-major firing order: ring fastest
    AZ, EL = np.meshgrid(az, elev, indexing="ij")
    RG = np.broadcast_to(rings[None, :], AZ.shape)
    CL = np.broadcast_to(cols[:, None], AZ.shape)
    ds = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], axis=-1).reshape(-1, 3)
    dw = ds @ R.T
    rng = _raycast(np.asarray(t, float), dw)
    g = np.random.default_rng(seed)
    noise = g.standard_normal(rng.shape[0]) * noise_sigma
    ok = np.isfinite(rng) & (rng < 100.0)
    r = (rng + noise)[ok]
    xyz = (ds[ok] * r[:, None]).astype(np.float32)
    ring = RG.reshape(-1)[ok].astype(np.uint16)
    col = CL.reshape(-1)[ok]
    tm = (col / H * 0.1).astype(np.float32)
    inten = (50.0 + 10.0 * np.cos(0.7 * r)).astype(np.float32)
    return Frame(xyz, inten, ring, tm, n_scan, H)


# nominal motion of SURVEY §8d
MOTION_RPY_DEG = (0.5, 1.0, 2.0)
MOTION_T = (0.30, 0.05, 0.02)
PREV_STEP_T = (0.28, 0.04, 0.02)


def make_pair(sensor: str, seed: int = SEED, rpy_deg=MOTION_RPY_DEG, t=MOTION_T, origin=None, **kw):
    """Frame k at identity and frame k+1 at the given pose. Returns (frame_k, frame_k1, (R, t)).
    origin = (x, y, yaw_deg): where in the hall frame k stands (default: the centre, facing +x) — the motion stays relative to frame k."""
    R = rpy_to_R(*np.deg2rad(rpy_deg))
    t = np.asarray(t, float)
    R0, t0 = np.eye(3), np.zeros(3)
    if origin is not None:
        R0 = rpy_to_R(0.0, 0.0, np.deg2rad(origin[2])); t0 = np.array([origin[0], origin[1], 0.0])
    f0 = make_frame(sensor, R0, t0, seed, **kw)
    f1 = make_frame(sensor, R0 @ R, t0 + R0 @ t, seed + 1, **kw)
    return f0, f1, (R, t)


def pool_origin(i: int):
    """Deterministic stand points for pools of DISTINCT frame pairs (bench.py): pair 0 is the nominal pair of SURVEY 8d at the centre of
    the hall, pair i > 0 stands somewhere else, facing elsewhere (different geometry in view, different noise)."""
    if i == 0:
        return None
    g = np.random.default_rng(SEED + 7919 * i)
    while True:   # keep 3 m clear of every obstacle's footprint
        x, y, yaw = float(g.uniform(-18.0, 18.0)), float(g.uniform(-10.0, 10.0)), float(g.uniform(-180.0, 180.0))
        clear = True
        for ob in OBSTACLES:
            if ob[0] == "box":
                dx, dy = max(abs(x - ob[1]) - ob[3], 0.0), max(abs(y - ob[2]) - ob[4], 0.0)
                clear &= (dx * dx + dy * dy) ** 0.5 >= 3.0
            else:
                clear &= ((x - ob[1]) ** 2 + (y - ob[2]) ** 2) ** 0.5 - ob[3] >= 3.0
        if clear:
            return (x, y, yaw)


def dense_pair(sensor: str, seed: int = SEED, **kw):
    """Dense-mode registration inputs (BASELINE configs 2-4): source = frame k pre-translated by the forward
    prediction (lidarOdometry.cpp:459,700-712), target = frame k+1; both (n,4) float32 x,y,z,intensity."""
    f0, f1, pose = make_pair(sensor, seed, **kw)
    pred = np.asarray(PREV_STEP_T, np.float32)
    # sensor moved by +t => scene points move by about -t in the sensor frame
    src = np.concatenate([f0.xyz - pred[None, :], f0.intensity[:, None]], axis=1).astype(np.float32)
    tgt = np.concatenate([f1.xyz, f1.intensity[:, None]], axis=1).astype(np.float32)
    return src, tgt, pose
