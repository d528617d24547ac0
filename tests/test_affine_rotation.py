"""CPU (host code of librolo_hip.so, no GPU needed): Eigen::Affine3f::rotation() as the odometry driver and TransformFusion read it
(src/lidarOdometry.cpp:130, 474, 548) — the float polar factor of the linear part, not the linear part (rolo_affine3f_rotation,
rolo_amd/csrc/polar_f32.hpp) — against the oracle's restatement (bit for bit: both restate Eigen's two-sided Jacobi sweeps in float),
against LAPACK's float SVD (a few float ulps) and against known answers."""
import ctypes as C

import numpy as np
from scipy.spatial.transform import Rotation

from oracle import pyorc, twin_odom
from rolo_amd import _lib

F = np.float32
fp = C.POINTER(C.c_float)


def hip_rotation(T):
    A = np.eye(4, dtype=F); T = np.asarray(T, F); A[:T.shape[0], :T.shape[1]] = T
    R = np.zeros((3, 3), F)
    _lib.lib().rolo_affine3f_rotation(A.ctypes.data_as(fp), R.ctypes.data_as(fp))
    return R


def chain_matrices(n, seed=11):
    """what transformation_interpolated looks like: products of float rotation matrices (getTransformation * getFinalTransformation)"""
    g = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        T = np.eye(3, dtype=F)
        for _ in range(int(g.integers(1, 4))):
            R = Rotation.from_euler("xyz", g.uniform(-0.2, 0.2, 3)).as_matrix().astype(F)
            T = (T @ R).astype(F)
        out.append(T)
    return out


def test_equals_the_oracle_bit_for_bit_and_lapack_to_float_ulps():
    differs = 0
    for T in chain_matrices(300):
        a = hip_rotation(T); b = pyorc.affine3f_rotation(T)
        assert np.array_equal(a, b)
        c = twin_odom.rotation_of_affine3f(np.pad(T, ((0, 1), (0, 1))))
        assert np.abs(a.astype(np.float64) - c.astype(np.float64)).max() < 1e-6
        # the polar factor is orthonormal to float rounding, and it is NOT the linear part: it moves the last ulps
        assert np.abs(a.astype(np.float64) @ a.astype(np.float64).T - np.eye(3)).max() < 1e-6
        assert np.abs(a - T).max() < 2e-6
        differs += int(not np.array_equal(a, T))
    assert differs > 100   # the case DESIGN.md used to wave away ("taken as the linear part") is the common one


def test_known_answers():
    R = Rotation.from_euler("xyz", [0.3, -0.2, 1.1]).as_matrix()
    S = np.array([[2.0, 0.3, 0.1], [0.3, 1.5, -0.2], [0.1, -0.2, 0.8]])      # symmetric positive definite: A = R S has polar factor R
    got = hip_rotation((R @ S).astype(F))
    assert np.abs(got - R).max() < 1e-6
    assert np.array_equal(hip_rotation(np.eye(3)), np.eye(3, dtype=F))
    # an improper linear part (reflection): Eigen flips the last singular direction so that det(rotation) = +1
    M = (R @ np.diag([1.0, 1.0, -1.0])).astype(F)
    got = hip_rotation(M)
    assert abs(np.linalg.det(got.astype(np.float64)) - 1.0) < 1e-6 and np.array_equal(got, pyorc.affine3f_rotation(M))
    # scaled rotation: the scale goes to the scaling factor, not the rotation
    assert np.abs(hip_rotation((3.5 * R).astype(F)) - R).max() < 1e-6
