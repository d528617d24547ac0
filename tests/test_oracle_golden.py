"""CPU: the C++ oracle (oracle/rolo_oracle.cpp) against the committed golden vectors produced by the independent
numpy/scipy twin (tests/golden/make_golden.py). This is the pin of the oracle (the reference's own tests hold
no vectors for this path — SURVEY.md §4)."""
import os

import numpy as np
import pytest

from oracle import pyorc

CASES = ["vlp16_polar", "os64_uniform", "vlp16_polar_fixed20", "vlp16_polar_q2"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def make_reg(g, threads=1):
    p = pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0), num_threads=threads,
                             voxel_type=int(g["voxel_type"]), voxel_resolution=float(g["leaf"]),
                             fixed_iterations=int(g["fixed_iterations"]), q2_intended=int(g["q2_intended"]))
    r = pyorc.Reg(p)
    r.set_target(g["target"])
    r.set_source(g["source"])
    return r


@pytest.mark.parametrize("name", CASES[:2])
def test_covariances_and_knn(golden_dir, name):
    g = load(golden_dir, name)
    r = make_reg(g)
    assert r.compute_covariances() == 0
    sub = g["cov_sub"]
    sc = r.source_covs()[sub][:, :3, :3]
    tc = r.target_covs()[sub][:, :3, :3]
    assert np.abs(sc - g["src_cov_sub"]).max() < 1e-9
    assert np.abs(tc - g["tgt_cov_sub"]).max() < 1e-9
    # 4th row/col of the Matrix4d are zero (rot_vgicp_impl.hpp:490)
    full = r.source_covs()
    assert np.all(full[:, 3, :] == 0) and np.all(full[:, :, 3] == 0)
    idx, d2 = pyorc.knn(g["source"], 20, 1)
    assert np.array_equal(idx[sub], g["src_knn_sub"])
    assert np.all(idx[:, 0] == np.arange(idx.shape[0]))  # query point itself first (d2 = 0)
    assert np.all(np.diff(d2, axis=1) >= 0)


@pytest.mark.parametrize("name", CASES[:2])
def test_voxel_keys_and_map_bit_exact(golden_dir, name):
    g = load(golden_dir, name)
    keys = pyorc.voxel_keys(g["target"], int(g["voxel_type"]), float(g["leaf"]))
    assert np.array_equal(keys, g["tgt_keys"].astype(np.int32))
    r = make_reg(g)
    assert r.build_voxelmap() == 0
    vk, vc, vm, vcov = r.voxels()
    assert np.array_equal(vk, g["vox_keys"].astype(np.int32))  # order of first appearance
    assert np.array_equal(vc, g["vox_count"])
    assert np.abs(vm[:, :3] - g["vox_mean"]).max() < 1e-12
    assert np.all(vm[:, 3] == 1.0)
    assert np.abs(vcov[::5, :3, :3] - g["vox_cov_sub"]).max() < 1e-9


@pytest.mark.parametrize("name", CASES[:2])
@pytest.mark.parametrize("threads", [1, 4])
def test_linearize_stages(golden_dir, name, threads):
    g = load(golden_dir, name)
    r = make_reg(g, threads)
    e, H, b = r.so3_linearize(g["T_probe"])
    s, v = r.correspondences()
    vk = r.voxels()[0]
    order = np.argsort(s, kind="stable")
    assert np.array_equal(s[order], g["corr_src"])  # bit-exact correspondence list
    assert np.array_equal(vk[v[order]], g["corr_vox_keys"].astype(np.int32))
    assert abs(e - g["so3_err"]) <= 1e-10 * abs(g["so3_err"])
    assert np.abs(H - g["so3_H"]).max() <= 1e-10 * np.abs(g["so3_H"]).max()
    assert np.abs(b - g["so3_b"]).max() <= 1e-10 * np.abs(g["so3_b"]).max()
    e2 = r.compute_error(g["T_probe2"])
    assert abs(e2 - g["err_probe2"]) <= 1e-10 * abs(g["err_probe2"])
    # translation-stage pieces use the correspondences cached by the last so3_linearize
    et, Ht, bt = r.t3_linearize(g["t_probe"], g["t_guess"], g["t_last"])
    assert abs(et - g["t3_err"]) <= 1e-10 * abs(g["t3_err"])
    assert np.abs(Ht - g["t3_H"]).max() <= 1e-10 * np.abs(g["t3_H"]).max()
    assert np.abs(bt - g["t3_b"]).max() <= 1e-10 * np.abs(g["t3_b"]).max()
    ev = r.compute_t_error(g["t_probe"], g["t_guess"], g["t_last"])
    assert abs(ev - g["t3_err_variant"]) <= 1e-10 * abs(g["t3_err_variant"])
    e6, H6, b6 = r.linearize(g["T_probe6"])
    assert abs(e6 - g["lin6_err"]) <= 1e-10 * abs(g["lin6_err"])
    assert np.abs(H6 - g["lin6_H"]).max() <= 1e-10 * np.abs(g["lin6_H"]).max()
    assert np.abs(b6 - g["lin6_b"]).max() <= 1e-10 * np.abs(g["lin6_b"]).max()


@pytest.mark.parametrize("name", CASES)
def test_full_solve(golden_dir, name):
    g = load(golden_dir, name)
    r = make_reg(g)
    rc, Tf, Td, it, conv = r.align()
    assert rc == 0
    assert it == int(g["align_iters"]) and conv == bool(g["align_converged"])
    assert np.abs(Td - g["align_T"]).max() < 1e-9
    assert np.abs(Tf - g["align_T"].astype(np.float32)).max() < 1e-6
    assert np.all(Td[:3, 3] == 0)  # SO3 stage is a pure rotation
    rc, t, tit = r.compute_translation(np.zeros(3), g["t_guess"], g["t_last"])
    assert rc == 0 and tit == int(g["trans_iters"])
    assert np.abs(t - g["trans_final"]).max() < 1e-9
    tr = r.trace()
    gt = g["trace"]
    # Once converged, y0 - yi is rounding noise and the accept / reject decision (sign of rho) is arbitrary, so
    # the LM trace is compared record by record only while the step is significant.
    for stage in (0, 1):
        ta = [a for a in tr if a["stage"] == stage]
        tb = [b for b in gt if int(b[0]) == stage]
        assert ta and tb
        for a, b in zip(ta, tb):
            if abs(b[4] - b[5]) <= 1e-7 * abs(b[4]):
                break
            assert (a["outer"], a["trial"], a["accepted"]) == tuple(int(x) for x in b[1:4])
            assert abs(a["y0"] - b[4]) <= 1e-9 * abs(b[4]) and abs(a["yi"] - b[5]) <= 1e-9 * abs(b[5])
            assert abs(a["lam"] - b[7]) <= 1e-6 * abs(b[7])
