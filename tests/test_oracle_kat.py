"""CPU: analytic known-answer tests for the oracle's hand-rolled pieces."""
import numpy as np
import pytest
from scipy.linalg import expm
from scipy.spatial.transform import Rotation

from oracle import pyorc
from rolo_amd import synth


def test_so3_exp_closed_form():
    rng = np.random.default_rng(1)
    for w in list(rng.normal(size=(20, 3)) * 0.3) + [np.zeros(3), np.array([1e-7, -2e-7, 3e-8]), np.array([3.0, 0.1, -0.2])]:
        R = pyorc.so3_exp(w)
        assert np.abs(R - Rotation.from_rotvec(w).as_matrix()).max() < 1e-13


def test_se3_exp_matches_matrix_exponential():
    rng = np.random.default_rng(2)
    for a in list(rng.normal(size=(20, 6)) * 0.2) + [np.zeros(6), np.array([0, 0, 0, 1.0, 2.0, 3.0])]:
        R, t = pyorc.se3_exp(a)
        X = np.zeros((4, 4))
        X[:3, :3] = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        X[:3, 3] = a[3:]
        E = expm(X)
        assert np.abs(R - E[:3, :3]).max() < 1e-12 and np.abs(t - E[:3, 3]).max() < 1e-12


def test_jacobi_svd_conventions():
    rng = np.random.default_rng(3)
    for _ in range(50):
        B = rng.normal(size=(3, 3))
        A = B @ B.T
        U, s, V = pyorc.svd3(A)
        assert np.all(np.diff(s) <= 0) and np.all(s >= 0)  # descending, non-negative
        assert np.abs(U @ np.diag(s) @ V.T - A).max() < 1e-12 * max(1, s[0])
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-13 and np.abs(V.T @ V - np.eye(3)).max() < 1e-13
        assert np.abs(s - np.linalg.svd(A, compute_uv=False)).max() < 1e-12 * max(1, s[0])
    # general (non-symmetric) input and a rank-deficient one
    A = rng.normal(size=(3, 3))
    U, s, V = pyorc.svd3(A)
    assert np.abs(U @ np.diag(s) @ V.T - A).max() < 1e-13
    n = np.array([0.0, 0.6, 0.8])
    P = np.eye(3) - np.outer(n, n)
    U, s, V = pyorc.svd3(P)
    assert abs(s[2]) < 1e-15 and abs(abs(U[:, 2] @ n) - 1) < 1e-12


def test_ldlt_solves_spd_systems():
    rng = np.random.default_rng(4)
    for n in (3, 6):
        for _ in range(20):
            B = rng.normal(size=(n, n)); A = B @ B.T + 0.1 * np.eye(n); b = rng.normal(size=n)
            rc, x = pyorc.ldlt_solve(A, b)
            assert rc == 0 and np.abs(x - np.linalg.solve(A, b)).max() < 1e-10


def test_polar_coord_hand_computed_bins():
    res = (0.175, 0.175, 2.0)
    # along -x: atan2(0,-1)+pi = 2pi -> floor(2pi/0.175) = 35 ; phi = pi/2 -> floor(8.97) = 8 ; r=5 -> 2
    pts = np.array([[-5.0, 0.0, 0.0], [5.0, 0.0, 0.0], [0.0, 0.0, 7.9], [3.0, 4.0, 0.0], [0.0, -4.1, 0.0]], np.float32)
    k = pyorc.voxel_keys(pts, pyorc.VOXEL_POLAR, 1.0, res)
    assert k[0].tolist() == [35, 8, 2]
    assert k[1].tolist() == [17, 8, 2]      # atan2(0,5)+pi = pi -> 17.95
    assert k[2].tolist() == [17, 0, 3]      # straight up: phi = 0
    assert k[3].tolist() == [int((np.arctan2(4, 3) + np.pi) / 0.175), 8, 2]
    assert k[4].tolist() == [8, 8, 2]       # atan2(-4.1,0)+pi = pi/2 -> 8.97
    # uniform: floor(x/leaf - 0.5)
    ku = pyorc.voxel_keys(np.array([[0.49, -0.51, 1.5], [0.5, 0.0, -1.49]], np.float32), pyorc.VOXEL_UNIFORM, 1.0)
    assert ku.tolist() == [[-1, -2, 1], [0, -1, -2]]


def test_mahalanobis_of_identity_pairs_is_half_identity():
    """cov_A = cov_B = I (NONE regularisation on a synthetic isotropic cloud is not available; use the algebra):
    (I + R I R^T)^-1 = I/2 — checked through so3_linearize on hand-made covariances is not exposed, so check the
    PLANE pair instead: both planes identical => RCR = 2 C, M = C^-1 / 2."""
    rng = np.random.default_rng(5)
    # a noise-free plane: every neighbourhood is exactly planar -> cov = U diag(1,1,1e-3) V^T with normal z
    xy = rng.uniform(-5, 5, size=(400, 2))
    pts = np.concatenate([xy, np.full((400, 1), -1.5), np.zeros((400, 1))], axis=1).astype(np.float32)
    p = pyorc.default_params(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=50.0, num_threads=1)
    r = pyorc.Reg(p)
    r.set_target(pts); r.set_source(pts)
    assert r.compute_covariances() == 0
    C = r.source_covs()[:, :3, :3]
    assert np.abs(C - np.diag([1, 1, 1e-3])).max() < 1e-9
    e, H, b = r.so3_linearize(np.eye(4))
    s, v, M = r.correspondences(with_mahalanobis=True)
    assert len(s) == 400  # one voxel holds everything
    assert np.abs(M[:, :3, :3] - np.diag([0.5, 0.5, 500.0])).max() < 1e-6
    assert np.all(M[:, 3, :] == 0) and np.all(M[:, :, 3] == 0)


def test_known_rotation_recovered_on_noise_free_scene():
    """Noise-free scene, pure small rotation, zero translation => the SO(3) stage recovers R to <= 1e-5 rad
    when iterated to convergence with a tight rotation epsilon."""
    f0 = synth.make_frame("vlp16", np.eye(3), np.zeros(3), 7, noise_sigma=0.0, col_stride=4)
    Rtrue = synth.rpy_to_R(*np.deg2rad([0.3, -0.2, 0.6]))
    src = np.concatenate([f0.xyz, np.zeros((f0.xyz.shape[0], 1), np.float32)], axis=1)
    tgt = src.copy()
    tgt[:, :3] = (f0.xyz.astype(np.float64) @ Rtrue.T).astype(np.float32)
    p = pyorc.default_params(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=1.0, num_threads=0, rotation_epsilon=1e-9,
                             max_iterations=64)
    r = pyorc.Reg(p)
    r.set_target(tgt); r.set_source(src)
    rc, Tf, Td, it, conv = r.align()
    assert rc == 0
    dR = Td[:3, :3] @ Rtrue.T
    ang = np.linalg.norm(Rotation.from_matrix(dR).as_rotvec())
    assert ang < 1e-5, ang


def test_too_few_points_is_an_error_not_ub():
    pts = np.random.default_rng(0).normal(size=(10, 4)).astype(np.float32)
    r = pyorc.Reg(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0)))
    r.set_target(pts); r.set_source(pts)
    assert r.compute_covariances() == -2


def test_deskew_known_answers():
    """ImageProjection::deskewPoint restated (imageProjection.cpp:368-396): zero increment = identity; at relTime = scanPeriod
    with odomTimeDiff = scanPeriod the stored point is the raw point rotated by -yaw about z; ranges / pixels are the raw ones."""
    fr = synth.make_frame("vlp16", np.eye(3), np.zeros(3), synth.SEED)
    fo = pyorc.front_params(n_scan=16, horizon_scan=1800)
    raw = pyorc.project(fo, fr.xyz, fr.ring)
    n = fr.xyz.shape[0]
    full = np.full(n, 0.1, np.float32)
    same = pyorc.project(fo, fr.xyz, fr.ring, full, pyorc.deskew([0, 0, 0], 0.1, 0.1))
    assert np.array_equal(same["extracted"], raw["extracted"])
    yaw = 0.05
    rot = pyorc.project(fo, fr.xyz, fr.ring, full, pyorc.deskew([0, 0, yaw], 0.1, 0.1))
    assert np.array_equal(rot["point_range"], raw["point_range"]) and np.array_equal(rot["point_col_ind"], raw["point_col_ind"])
    c, s_ = np.cos(-yaw), np.sin(-yaw)
    want = raw["extracted"][:, :3].astype(np.float64) @ np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]).T
    assert np.abs(rot["extracted"][:, :3] - want).max() < 2e-5
    assert np.array_equal(rot["extracted"][:, 3], raw["extracted"][:, 3])
    half = pyorc.project(fo, fr.xyz, fr.ring, np.full(n, 0.05, np.float32), pyorc.deskew([0, 0, yaw], 0.1, 0.1))
    c, s_ = np.cos(-yaw / 2), np.sin(-yaw / 2)
    want = raw["extracted"][:, :3].astype(np.float64) @ np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]]).T
    assert np.abs(half["extracted"][:, :3] - want).max() < 2e-5
    inc = pyorc.odom_increment([1.0, 2.0, 0.5, 0.0, 0.0, 0.3], [1.0, 2.0, 0.5, 0.0, 0.0, 0.35])
    assert np.abs(inc - np.array([0, 0, 0, 0, 0, 0.05])).max() < 1e-6
