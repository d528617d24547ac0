"""CPU: the options of RotVGICP the reference never selects (other regularisations, DIRECT7 / DIRECT27) in the C++ oracle
against their independent numpy statement (oracle/twin_options.py)."""
import numpy as np
import pytest

from oracle import pyorc, twin_options as topt
from oracle.twin import Twin
from rolo_amd import synth


@pytest.fixture(scope="module")
def pair():
    src, tgt, _ = synth.dense_pair("vlp16", col_stride=8)
    return src, tgt


@pytest.mark.parametrize("method", [topt.NONE, topt.MIN_EIG, topt.NORMALIZED_MIN_EIG, topt.PLANE, topt.FROBENIUS, topt.PLANE_S])
def test_regularization_methods(pair, method):
    src, tgt = pair
    o = pyorc.Reg(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0), regularization=method))
    o.set_target(tgt); o.set_source(src)
    assert o.compute_covariances() == 0
    want = topt.covariances(src[:, :3].astype(np.float32), method)
    got = o.source_covs()[:, :3, :3]
    scale = np.abs(want).max(axis=(1, 2), keepdims=True)
    # FROBENIUS inverts twice: conditioning costs a few digits
    tol = 1e-6 if method == topt.FROBENIUS else 1e-9
    assert np.abs(got - want).max() <= tol * max(1.0, float(scale.max())) or np.abs((got - want) / scale).max() <= tol
    assert np.all(o.source_covs()[:, 3, :] == 0) and np.all(o.source_covs()[:, :, 3] == 0)   # 4x4 with a zero last row / column


@pytest.mark.parametrize("neighbor,n_off", [(1, 7), (0, 27)])
def test_multi_offset_correspondences_and_linearisation(pair, neighbor, n_off):
    src, tgt = pair
    o = pyorc.Reg(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0), neighbor_search=neighbor))
    o.set_target(tgt); o.set_source(src)
    tw = Twin(src, tgt, voxel_type="polar", polar_res=(0.175, 0.175, 2.0))
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.004, -0.007, 0.02)
    e, H, b = o.so3_linearize(T)
    et, Ht, bt, src_i, vox_keys = topt.so3_linearize_multi(tw, T, n_off)
    s, v = o.correspondences()
    keys_o, _, _, _ = o.voxels()
    # same pairs; their order is the OpenMP thread order in the reference (per-thread vectors, schedule(guided, 8)): unspecified
    def canon(si, kk):
        a = np.concatenate([np.asarray(si, np.int64)[:, None], np.asarray(kk, np.int64)], axis=1)
        return a[np.lexsort(a.T[::-1])]
    assert np.array_equal(canon(s, keys_o[v]), canon(src_i, vox_keys))
    assert abs(e - et) <= 1e-9 * abs(et) and np.abs(H - Ht).max() <= 1e-9 * np.abs(Ht).max() and np.abs(b - bt).max() <= 1e-9 * np.abs(bt).max()


@pytest.mark.parametrize("optimizer,name", [(1, "lm"), (0, "gn")])
def test_six_dof_drivers(pair, optimizer, name):
    """computeTransformation with step_lm / step_gn (lsq_registration_impl.hpp:152-270) against the twin's restatement."""
    src, tgt = pair
    o = pyorc.Reg(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0), optimizer=optimizer))
    o.set_target(tgt); o.set_source(src)
    rc, Tf, Td, iters, conv = o.align()
    tw = Twin(src, tgt, voxel_type="polar", polar_res=(0.175, 0.175, 2.0))
    x0, it, cv = topt.align6(tw, name)
    assert rc == 0 and iters == it and bool(conv) == bool(cv)
    assert np.abs(Td - x0).max() <= 1e-8
