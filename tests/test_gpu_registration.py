"""GPU parity: the HIP registration path (through the C ABI) against the oracle on the same seeded inputs, and
against the committed golden fixtures. Bars (BASELINE.json north_star): voxel / correspondence indices
bit-exact; rotation <= 1e-5 rad, translation <= 1e-4 m (observed: ~1e-10)."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import pyorc
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP, VoxelType, NeighborSearchMethod, LSQ_OPTIMIZER_TYPE, RegularizationMethod

pytestmark = pytest.mark.gpu

G = -np.asarray(synth.PREV_STEP_T)
L0 = G * 0.97


def rot_angle(Ra, Rb):
    return float(np.linalg.norm(Rotation.from_matrix(Ra @ Rb.T).as_rotvec()))


def make_pair(kind):
    if kind == "vlp16_polar":
        src, tgt, _ = synth.dense_pair("vlp16", col_stride=2)
        return src, tgt, dict(voxel_type=0, polar=(0.175, 0.175, 2.0), leaf=1.0)
    if kind == "os64_uniform":
        src, tgt, _ = synth.dense_pair("os1-64", col_stride=4)
        return src, tgt, dict(voxel_type=1, polar=(0.175, 0.175, 2.0), leaf=1.0)
    raise KeyError(kind)


def make_both(src, tgt, cfg, fixed=0, q2=0, optimizer=2, neighbor=2, threads=0):
    p = pyorc.default_params(polar_resolution=cfg["polar"], voxel_type=cfg["voxel_type"], voxel_resolution=cfg["leaf"],
                             fixed_iterations=fixed, q2_intended=q2, optimizer=optimizer, neighbor_search=neighbor,
                             num_threads=threads)
    o = pyorc.Reg(p)
    o.set_target(tgt); o.set_source(src)
    g = RotVGICP()
    if cfg["voxel_type"] == 0:
        g.setPolarResolution(*cfg["polar"])
    else:
        g.setResolution(cfg["leaf"])
    g.setFixedIterations(fixed); g.setQ2Intended(bool(q2)); g.setOptimizerType(optimizer); g.setNeighborSearchMethod(neighbor)
    g.setInputTarget(tgt); g.setInputSource(src)
    return o, g


@pytest.fixture(scope="module", params=["vlp16_polar", "os64_uniform"])
def pair(request):
    src, tgt, cfg = make_pair(request.param)
    return request.param, src, tgt, cfg


def test_knn_lists_bit_exact(pair):
    _, src, tgt, cfg = pair
    o, g = make_both(src, tgt, cfg)
    idx_o, d2_o = pyorc.knn(src, 20)
    idx_g, d2_g = g.knn(0)
    assert np.array_equal(idx_g, idx_o)
    assert np.array_equal(d2_g, d2_o)  # float32 distances bit-identical (no FMA contraction on either side)


def test_covariances(pair):
    _, src, tgt, cfg = pair
    o, g = make_both(src, tgt, cfg)
    assert o.compute_covariances() == 0
    g.computeCovariances()
    for co, cg in ((o.source_covs(), g.getSourceCovariances()), (o.target_covs(), g.getTargetCovariances())):
        assert np.abs(cg - co).max() < 1e-9
        assert np.all(cg[:, 3, :] == 0) and np.all(cg[:, :, 3] == 0)


@pytest.mark.parametrize("reg", [RegularizationMethod.NONE, RegularizationMethod.MIN_EIG, RegularizationMethod.NORMALIZED_MIN_EIG,
                                 RegularizationMethod.FROBENIUS, RegularizationMethod.PLANE_S])
def test_other_regularizations(reg):
    src, tgt, cfg = make_pair("vlp16_polar")
    p = pyorc.default_params(polar_resolution=cfg["polar"], regularization=reg)
    o = pyorc.Reg(p); o.set_target(tgt[:4000]); o.set_source(src[:4000])
    assert o.compute_covariances() == 0
    g = RotVGICP(); g.setPolarResolution(*cfg["polar"]); g.setRegularizationMethod(reg)
    g.setInputTarget(tgt[:4000]); g.setInputSource(src[:4000]); g.computeCovariances()
    co, cg = o.source_covs(), g.getSourceCovariances()
    assert np.abs(cg - co).max() <= 1e-9 * max(1.0, np.abs(co).max())


def test_voxel_keys_and_map(pair):
    _, src, tgt, cfg = pair
    o, g = make_both(src, tgt, cfg)
    keys_o = pyorc.voxel_keys(tgt, cfg["voxel_type"], cfg["leaf"], cfg["polar"])
    assert np.array_equal(g.targetVoxelKeys(), keys_o)  # bit-exact voxel indices
    assert o.build_voxelmap() == 0
    g.buildVoxelMap()
    ko, co, mo, vo = o.voxels()
    kg, cg, mg, vg = g.voxels()
    assert kg.shape == ko.shape
    so = np.lexsort(ko.T[::-1]); sg = np.lexsort(kg.T[::-1])
    assert np.array_equal(kg[sg], ko[so])
    assert np.array_equal(cg[sg], co[so])
    assert np.abs(mg[sg] - mo[so]).max() < 1e-12   # fixed-point sums of float coordinates are exact; the oracle's serial fp64 sum rounds
    assert np.abs(vg[sg] - vo[so]).max() < 1e-9
    # the bin-edge guard: target points within 1e-12 (in bins) of a POLAR bin edge are the only ones device libm could have put into
    # another voxel than glibc does. The synthetic scans do have a few (the azimuth wrap at -pi: atan2 + pi ~ 1e-16); the library counts
    # them, the count agrees with the same criterion evaluated with glibc, and their keys (checked above for every point) still match.
    if cfg["voxel_type"] == 0:
        x, y, z = (tgt[:, k].astype(np.float64) for k in range(3)); r = np.sqrt((x * x + y * y) + z * z)
        q = np.c_[(np.arctan2(y, x) + np.pi) / cfg["polar"][0], np.arccos(z / r) / cfg["polar"][1], r / cfg["polar"][2]]
        fr = q - np.floor(q)
        assert g.numEdgePoints() == int(np.any((fr < 1e-12) | (fr > 1 - 1e-12), axis=1).sum()) <= 2 * 128
    else:
        assert g.numEdgePoints() == 0   # UNIFORM keys are correctly rounded on both sides: nothing to guard


def test_voxel_map_is_bit_reproducible(pair):
    """K6 accumulates in 64-bit fixed point (integer atomics): every voxel's count, mean and covariance are the same BITS run after
    run, in a fresh context, and whichever order the points arrive in (SURVEY 8e asked for deterministic sums)."""
    _, src, tgt, cfg = pair
    recs = []
    for trial in range(3):
        _, g = make_both(src, tgt, cfg)
        g.buildVoxelMap()
        if trial == 2:
            g.buildVoxelMap()   # rebuilt in the same context: the Hilbert-ordered build may be picked now (points per voxel known)
        k, c, m, v = g.voxels()
        o = np.lexsort(k.T[::-1])
        recs.append((k[o], c[o], m[o], v[o]))
        g.close()
    for r in recs[1:]:
        for a, b in zip(recs[0], r):
            assert np.array_equal(a, b)
    # a permuted target gives the same map as well (sums do not depend on the order of the addends) once its covariances are the same bits
    perm = np.random.default_rng(1).permutation(tgt.shape[0])
    _, g0 = make_both(src, tgt, cfg); g0.computeCovariances(); cov = g0.getTargetCovariances()
    _, g1 = make_both(src, tgt[perm], cfg); g1.computeCovariances()
    assert np.array_equal(g1.getTargetCovariances(), cov[perm])   # exact kNN: covariances do not depend on the input order either
    g1.buildVoxelMap()
    k, c, m, v = g1.voxels(); o = np.lexsort(k.T[::-1])
    assert np.array_equal(k[o], recs[0][0]) and np.array_equal(c[o], recs[0][1]) and np.array_equal(m[o], recs[0][2]) and np.array_equal(v[o], recs[0][3])


def _register_once(kind):
    src, tgt, cfg = make_pair(kind)
    _, g = make_both(src, tgt, cfg)
    z = np.zeros(3)
    g.register_async(None, z, z, z, 0.1, 0.1, 0.3)
    Tf, Td, t = g.register_wait()
    k, c, m, v = g.voxels(); o = np.lexsort(k.T[::-1])
    g.close()
    return np.asarray(Td, np.float64), np.asarray(t, np.float64), (k[o], c[o], m[o], v[o])


def test_voxel_map_built_inside_the_search_is_the_same_map(pair):
    """rolo_register_async builds the target's voxel map inside the search's launches (VoxelFuse: cleared beside the key kernel, filled beside
    the sort scatters, accumulated by the covariance tail in curve order); rolo_build_voxelmap builds it with its own launches in input
    order. Integer sums: the two maps are the same bits, and so is a registration with the fusion switched off (a fresh process: the
    switch is read once)."""
    kind, src, tgt, cfg = pair
    _, g = make_both(src, tgt, cfg)
    g.buildVoxelMap()
    k, c, m, v = g.voxels(); o = np.lexsort(k.T[::-1])
    ref = (k[o], c[o], m[o], v[o])
    g.close()
    Td, t, fused = _register_once(kind)
    for a, b in zip(ref, fused):
        assert np.array_equal(a, b)
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); from tests.test_gpu_registration import _register_once;"
            "Td, t, _ = _register_once(%r); print(json.dumps([Td.ravel().tolist(), t.tolist()]))" % (root, kind))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, ROLO_VOXEL_FUSE="0"), cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    Td0, t0 = json.loads(r.stdout.strip().splitlines()[-1])
    assert np.array_equal(Td.ravel(), np.asarray(Td0)) and np.array_equal(t, np.asarray(t0))


@pytest.mark.parametrize("n", [21, 65, 257, 1025, 4097, 16385, 30011])
def test_odd_sizes_through_the_sort_tree_and_fused_map(n):
    """sizes around the sort's tile / workgroup / leaf boundaries: neighbour lists bit-exact against the oracle (small n), covariances and
    voxel map of the fused frame path (twice: eager, then a replayed graph) the same bits as the separate launches"""
    rng = np.random.default_rng(n)
    src0, tgt0, _ = synth.dense_pair("os1-64", col_stride=2)
    si = np.sort(rng.choice(src0.shape[0], n, replace=False)); ti = np.sort(rng.choice(tgt0.shape[0], max(21, n - int(rng.integers(0, 7))), replace=False))
    src = np.ascontiguousarray(src0[si]); tgt = np.ascontiguousarray(tgt0[ti])
    def ctx():
        g = RotVGICP(); g.setResolution(1.0); g.setInputTarget(tgt); g.setInputSource(src)
        return g
    if n <= 4097:
        idx_o, d2_o = pyorc.knn(src, 20)
        g = ctx(); idx_g, d2_g = g.knn(0); g.close()
        assert np.array_equal(idx_o, idx_g) and np.array_equal(d2_o, d2_g)
    g = ctx(); g.buildVoxelMap()
    k, c, m, v = g.voxels(); o = np.lexsort(k.T[::-1]); ref = (k[o], c[o], m[o], v[o]); cov_ref = g.getTargetCovariances().copy()
    g.close()
    g = ctx(); z = np.zeros(3)
    for it in range(3):
        if it:
            g.setInputTarget(tgt); g.setInputSource(src)
        g.register_async(None, z, z, z, 0.1, 0.1, 0.3)
        try:
            g.register_wait()
        except Exception as ex:   # a handful of scattered points: an empty correspondence set is the reference's behaviour too (ROLO_ENOCORR)
            assert "-4" in str(ex) and n < 1000
        k, c, m, v = g.voxels(); o = np.lexsort(k.T[::-1])
        for a, b in zip(ref, (k[o], c[o], m[o], v[o])):
            assert np.array_equal(a, b)
        assert np.array_equal(cov_ref, g.getTargetCovariances())
    g.close()


def _degenerate_cloud(kind, n, rng):
    """geometry the synthetic hall never produces: exact distance ties, duplicated points, everything on a line / in a plane, far-apart clusters"""
    if kind == "lattice":       # integer lattice: most of a point's neighbours are at exactly equal squared distances (ties broken by index)
        m = int(np.ceil(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        pts = g[rng.permutation(g.shape[0])[:n]] * np.float32(0.25)
    elif kind == "duplicates":  # every point three times (zero distances, identical keys but for the index)
        base = rng.uniform(-20, 20, (n // 3 + 1, 3)).astype(np.float32)
        pts = np.concatenate([base, base, base])[rng.permutation(3 * base.shape[0])[:n]]
    elif kind == "line":        # rank-1 neighbourhoods: two singular values of every covariance are zero
        t = np.sort(rng.uniform(-30, 30, n)).astype(np.float32)
        pts = np.stack([t, np.float32(0.5) * t + np.float32(1.0), np.float32(-0.25) * t], 1)
    elif kind == "plane":       # rank-2 neighbourhoods
        uv = rng.uniform(-15, 15, (n, 2)).astype(np.float32)
        pts = np.stack([uv[:, 0], uv[:, 1], np.float32(0.1) * uv[:, 0] - np.float32(0.2) * uv[:, 1]], 1)
    else:                       # "clusters": tight blobs 60 m apart — the curve's packets straddle empty space
        c = rng.uniform(-60, 60, (7, 3)).astype(np.float32)
        pts = (c[rng.integers(0, 7, n)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([pts.astype(np.float32), np.ones((pts.shape[0], 1), np.float32)], 1))


@pytest.mark.parametrize("kind", ["lattice", "duplicates", "line", "plane", "clusters"])
@pytest.mark.parametrize("n,k", [(333, 20), (2050, 20), (1500, 7), (1200, 33)])
def test_degenerate_geometry_matches_the_oracle(kind, n, k):
    """exact ties, duplicates, rank-deficient neighbourhoods, clusters far apart: neighbour lists and float distances bit-exact (ties by index,
    as the oracle orders them), covariances after the PLANE regularisation within 1e-9, voxel keys identical"""
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind}-{n}-{k}".encode()))
    src = _degenerate_cloud(kind, n, rng)
    tgt = _degenerate_cloud(kind, n + 17, rng)
    p = pyorc.default_params(voxel_type=1, voxel_resolution=1.0, k_correspondences=k)
    o = pyorc.Reg(p); o.set_target(tgt); o.set_source(src)
    g = RotVGICP(); g.setResolution(1.0); g.setCorrespondenceRandomness(k); g.setInputTarget(tgt); g.setInputSource(src)
    for which, cloud in ((0, src), (1, tgt)):
        idx_o, d2_o = pyorc.knn(cloud, k)
        idx_g, d2_g = g.knn(which)
        assert np.array_equal(d2_g, d2_o)
        assert np.array_equal(idx_g, idx_o)
    assert o.compute_covariances() == 0
    g.computeCovariances()
    for co, cg in ((o.source_covs(), g.getSourceCovariances()), (o.target_covs(), g.getTargetCovariances())):
        assert np.isfinite(cg).all()
        if kind == "line":
            # two singular values of the neighbourhood covariance are exactly zero: the SVD's U and V are then free to differ inside the null space
            # (sign, rotation — decided by rounding noise, in Eigen as much as here), so U diag(1, 1, 1e-3) V^T is well defined only along the
            # line. The oracle's matrix is no yardstick for the rest; what both must give: finite entries and the line direction kept with weight 1
            d = np.array([1.0, 0.5, -0.25]); d /= np.linalg.norm(d)
            for c in (co, cg):
                assert np.abs(c[:, :3, :3] @ d - d).max() <= 1e-6
        else:
            assert np.abs(cg - co).max() <= 1e-9 * max(1.0, np.abs(co).max())
    g.buildVoxelMap()
    kk = g.voxels()[0]
    ko = np.unique(pyorc.voxel_keys(tgt, 1, 1.0), axis=0)
    assert np.array_equal(kk[np.lexsort(kk.T[::-1])], ko[np.lexsort(ko.T[::-1])])
    g.close()


@pytest.mark.parametrize("k,overlap", [(10, True), (27, False), (32, True)])
def test_fused_map_with_other_k_and_search_modes(k, overlap):
    """the map built inside the search's launches for k != 20 (the 32-slot kernels) and with one search chain per cloud: same bits as
    the separate launches, also when a second registration finds the covariances already there"""
    src, tgt, _ = synth.dense_pair("os1-64", col_stride=4)
    def ctx():
        g = RotVGICP(); g.setResolution(1.0); g.setCorrespondenceRandomness(k); g.setOverlapKnn(overlap); g.setInputTarget(tgt); g.setInputSource(src)
        return g
    g = ctx(); g.buildVoxelMap()
    kk, c, m, v = g.voxels(); o = np.lexsort(kk.T[::-1]); ref = (kk[o], c[o], m[o], v[o]); cov = g.getTargetCovariances().copy()
    g.close()
    g = ctx(); z = np.zeros(3)
    for it in range(3):
        if it == 1:
            g.setInputTarget(tgt); g.setInputSource(src)   # it == 2: same clouds again, covariances kept
        g.register_async(None, z, z, z, 0.1, 0.1, 0.3); g.register_wait()
        kk, c, m, v = g.voxels(); o = np.lexsort(kk.T[::-1])
        for a, b in zip(ref, (kk[o], c[o], m[o], v[o])):
            assert np.array_equal(a, b)
        assert np.array_equal(cov, g.getTargetCovariances())
    g.close()


@pytest.mark.parametrize("k", [48, 64, 65, 100, 130])
def test_many_neighbours(k):
    """k above 32 (the reference takes any k): the 64-slot kernels, and above 64 the rounds of 64 (each round the 64 nearest beyond the previous
    round's last key) — neighbour lists bit-exact against the oracle, covariances and pose agree"""
    src, tgt, cfg = make_pair("os64_uniform")
    p = pyorc.default_params(polar_resolution=cfg["polar"], voxel_type=cfg["voxel_type"], voxel_resolution=cfg["leaf"], k_correspondences=k)
    o = pyorc.Reg(p); o.set_target(tgt); o.set_source(src)
    g = RotVGICP(); g.setResolution(cfg["leaf"]); g.setCorrespondenceRandomness(k); g.setInputTarget(tgt); g.setInputSource(src)
    idx_o, d2_o = pyorc.knn(src, k)
    idx_g, d2_g = g.knn(0)
    assert np.array_equal(idx_g, idx_o) and np.array_equal(d2_g, d2_o)
    assert o.compute_covariances() == 0
    g.computeCovariances()
    for co, cg in ((o.source_covs(), g.getSourceCovariances()), (o.target_covs(), g.getTargetCovariances())):
        assert np.abs(cg - co).max() < 1e-9
    rc, _, Td_o, _, _ = o.align(None)
    assert rc == 0
    g.align(None)
    assert rot_angle(np.asarray(g.final_transformation_d)[:3, :3], Td_o[:3, :3]) <= 1e-5
    g.close()


def test_linearize_stages(pair):
    _, src, tgt, cfg = pair
    o, g = make_both(src, tgt, cfg)
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.004, -0.007, 0.02)
    eo, Ho, bo = o.so3_linearize(T)
    eg, Hg, bg = g.so3_linearize(T)
    assert abs(eg - eo) <= 1e-9 * abs(eo)
    assert np.abs(Hg - Ho).max() <= 1e-9 * np.abs(Ho).max()
    assert np.abs(bg - bo).max() <= 1e-9 * np.abs(bo).max()
    # correspondence list bit-exact: same source points, same voxel keys
    s, v = o.correspondences()
    vk = o.voxels()[0]
    found, keys = g.correspondences()
    assert np.array_equal(np.nonzero(found[:, 0])[0], np.sort(s))
    order = np.argsort(s, kind="stable")
    assert np.array_equal(keys[s[order], 0], vk[v[order]])
    T2 = np.eye(4); T2[:3, :3] = synth.rpy_to_R(0.0045, -0.0065, 0.021)
    e2o, e2g = o.compute_error(T2), g.compute_error(T2)
    assert abs(e2g - e2o) <= 1e-9 * abs(e2o)
    tp = np.array([0.01, -0.004, 0.002])
    eo, Ho, bo = o.t3_linearize(tp, G, L0)
    eg, Hg, bg = g.t3_linearize(tp, G, L0)
    assert abs(eg - eo) <= 1e-9 * abs(eo)
    assert np.abs(Hg - Ho).max() <= 1e-9 * np.abs(Ho).max()
    assert np.abs(bg - bo).max() <= 1e-9 * np.abs(bo).max()
    e3o, e3g = o.compute_t_error(tp, G, L0), g.compute_t_error(tp, G, L0)
    assert abs(e3g - e3o) <= 1e-9 * abs(e3o)
    T6 = T.copy(); T6[:3, 3] = (0.01, -0.02, 0.005)
    eo, Ho, bo = o.linearize(T6)
    eg, Hg, bg = g.linearize(T6)
    assert abs(eg - eo) <= 1e-9 * abs(eo)
    assert np.abs(Hg - Ho).max() <= 1e-9 * np.abs(Ho).max()
    assert np.abs(bg - bo).max() <= 1e-9 * np.abs(bo).max()


def check_solve(o, g, guess=None):
    rc, Tf_o, Td_o, it_o, cv_o = o.align(guess)
    assert rc == 0
    Tf_g = g.align(guess)
    Td_g = g.final_transformation_d
    assert rot_angle(Td_g[:3, :3], Td_o[:3, :3]) <= 1e-5  # the bar
    assert np.abs(Td_g - Td_o).max() < 1e-8             # what we actually get
    assert np.abs(Tf_g - Tf_o).max() < 1e-6
    st = g.last_stats
    assert st.n_outer == it_o and bool(st.converged) == cv_o and st.lm_failed == 0
    rc, t_o, tit_o = o.compute_translation(np.zeros(3), G, L0)
    assert rc == 0
    t_g = g.computeTranslation(np.zeros(3), G, L0)
    assert np.abs(t_g - t_o).max() <= 1e-4
    assert np.abs(t_g - t_o).max() < 1e-8
    assert g.last_translation_stats.n_outer == tit_o
    # LM trace: same decisions while the step is significant
    tr_o, tr_g = o.trace(), g.trace()
    for stage in (0, 1):
        a = [r for r in tr_o if r["stage"] == stage]; b = [r for r in tr_g if r["stage"] == stage]
        assert a and b
        for ro, rg in zip(a, b):
            if abs(ro["y0"] - ro["yi"]) <= 1e-7 * abs(ro["y0"]):
                break
            assert (ro["outer"], ro["trial"], ro["accepted"]) == (rg["outer"], rg["trial"], rg["accepted"])
            assert abs(ro["y0"] - rg["y0"]) <= 1e-9 * abs(ro["y0"]) and abs(ro["yi"] - rg["yi"]) <= 1e-9 * abs(ro["yi"])
            assert abs(ro["lam"] - rg["lam"]) <= 1e-6 * abs(ro["lam"])
    return Td_g, t_g


@pytest.mark.parametrize("fixed,q2", [(0, 0), (20, 0), (0, 1)])
def test_full_solve_matches_oracle(pair, fixed, q2):
    _, src, tgt, cfg = pair
    o, g = make_both(src, tgt, cfg, fixed=fixed, q2=q2)
    check_solve(o, g)


def test_full_solve_with_guess(pair):
    _, src, tgt, cfg = pair
    o, g = make_both(src, tgt, cfg)
    guess = np.eye(4, dtype=np.float32)
    guess[:3, :3] = synth.rpy_to_R(0.002, 0.001, 0.01)
    guess[:3, 3] = (0.02, -0.01, 0.0)
    check_solve(o, g, guess)


@pytest.mark.parametrize("optimizer", [LSQ_OPTIMIZER_TYPE.LevenbergMarquardt, LSQ_OPTIMIZER_TYPE.GaussNewton])
def test_six_dof_optimizers(optimizer):
    src, tgt, cfg = make_pair("os64_uniform")
    o, g = make_both(src, tgt, cfg, optimizer=optimizer)
    rc, Tf_o, Td_o, it_o, cv_o = o.align()
    Tf_g = g.align()
    Td_g = g.final_transformation_d
    assert rot_angle(Td_g[:3, :3], Td_o[:3, :3]) <= 1e-5 and np.abs(Td_g[:3, 3] - Td_o[:3, 3]).max() <= 1e-4
    assert np.abs(Td_g - Td_o).max() < 1e-7
    assert g.last_stats.n_outer == it_o


@pytest.mark.parametrize("neighbor", [NeighborSearchMethod.DIRECT7, NeighborSearchMethod.DIRECT27])
def test_multi_offset_neighbor_search(neighbor):
    src, tgt, cfg = make_pair("os64_uniform")
    o, g = make_both(src[:6000], tgt[:6000], cfg, neighbor=neighbor)
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.004, -0.007, 0.02)
    eo, Ho, bo = o.so3_linearize(T)
    eg, Hg, bg = g.so3_linearize(T)
    assert o.correspondences()[0].shape[0] == int(g.correspondences()[0].sum())
    assert abs(eg - eo) <= 1e-9 * abs(eo) and np.abs(Hg - Ho).max() <= 1e-9 * np.abs(Ho).max()
    rc, _, Td_o, it_o, _ = o.align()
    g.align()
    assert np.abs(g.final_transformation_d - Td_o).max() < 1e-8


@pytest.mark.parametrize("name", ["vlp16_polar", "os64_uniform", "vlp16_polar_fixed20", "vlp16_polar_q2"])
def test_against_committed_golden(golden_dir, name):
    gd = np.load(os.path.join(golden_dir, name + ".npz"))
    g = RotVGICP()
    if int(gd["voxel_type"]) == 0:
        g.setPolarResolution(0.175, 0.175, 2.0)
    else:
        g.setResolution(float(gd["leaf"]))
    g.setFixedIterations(int(gd["fixed_iterations"])); g.setQ2Intended(bool(gd["q2_intended"]))
    g.setInputTarget(gd["target"]); g.setInputSource(gd["source"])
    assert np.array_equal(g.targetVoxelKeys(), gd["tgt_keys"].astype(np.int32))
    g.computeCovariances()
    sub = gd["cov_sub"]
    assert np.abs(g.getSourceCovariances()[sub][:, :3, :3] - gd["src_cov_sub"]).max() < 1e-9
    assert np.array_equal(g.knn(0)[0][sub], gd["src_knn_sub"])
    e, H, b = g.so3_linearize(gd["T_probe"])
    assert abs(e - gd["so3_err"]) <= 1e-9 * abs(gd["so3_err"])
    assert np.abs(H - gd["so3_H"]).max() <= 1e-9 * np.abs(gd["so3_H"]).max()
    found, keys = g.correspondences()
    assert np.array_equal(np.nonzero(found[:, 0])[0], gd["corr_src"])
    assert np.array_equal(keys[gd["corr_src"], 0], gd["corr_vox_keys"].astype(np.int32))
    g.align()
    assert g.last_stats.n_outer == int(gd["align_iters"])
    assert rot_angle(g.final_transformation_d[:3, :3], gd["align_T"][:3, :3]) <= 1e-5
    assert np.abs(g.final_transformation_d - gd["align_T"]).max() < 1e-8
    t = g.computeTranslation(np.zeros(3), gd["t_guess"], gd["t_last"])
    assert np.abs(t - gd["trans_final"]).max() <= 1e-4
    assert g.last_translation_stats.n_outer == int(gd["trans_iters"])


def test_async_register_equals_two_calls(pair):
    _, src, tgt, cfg = pair
    _, g1 = make_both(src, tgt, cfg, fixed=20)
    g1.align(); T1 = g1.final_transformation_d.copy()
    t1 = g1.computeTranslation(np.zeros(3), G, L0)
    _, g2 = make_both(src, tgt, cfg, fixed=20)
    g2.register_async(None, np.zeros(3), G, L0)
    Tf, Td, t2 = g2.register_wait()
    # same kernels in the same order (the voxel sums are order-independent 64-bit integers: nothing in the two runs may differ beyond rounding)
    assert np.abs(Td - T1).max() < 1e-12 and np.abs(t2 - t1).max() < 1e-12
    assert g2.last_stats.n_outer == 20


def test_errors_are_codes_not_crashes():
    from rolo_amd._lib import RoloError
    g = RotVGICP(); g.setPolarResolution(0.175, 0.175, 2.0)
    pts = np.random.default_rng(0).normal(size=(10, 4)).astype(np.float32) * 5
    g.setInputTarget(pts); g.setInputSource(pts)
    with pytest.raises(RoloError) as ei:
        g.align()
    assert ei.value.code == -2  # ROLO_ETOOFEW
    g2 = RotVGICP(); g2.setPolarResolution(0.175, 0.175, 2.0)
    src, tgt, cfg = make_pair("vlp16_polar")
    far = src.copy(); far[:, :3] *= 40.0  # no source point falls into an occupied target voxel
    g2.setInputTarget(tgt); g2.setInputSource(far)
    with pytest.raises(RoloError) as ei:
        g2.align()
    assert ei.value.code == -4  # ROLO_ENOCORR


def test_state_errors_of_the_async_batch_and_handover_entry_points():
    """Misuse returns ROLO_ESTATE / ROLO_EUNSUPPORTED with a message; nothing is left half-enqueued."""
    from rolo_amd._lib import RoloError
    from rolo_amd.rotvgicp import RotVGICPBatch
    src, tgt, cfg = make_pair("vlp16_polar")
    g = RotVGICP(); g.setPolarResolution(*cfg["polar"])
    with pytest.raises(RoloError) as ei:
        g.register_wait()                      # nothing in flight
    assert ei.value.code == -5
    g.setInputTarget(tgt); g.setInputSource(src)
    with pytest.raises(RoloError) as ei:
        g.adoptTargetCovariances()             # the target has no covariances yet
    assert ei.value.code == -5
    g.register_async(None, np.zeros(3), G, L0)
    with pytest.raises(RoloError) as ei:
        g.register_async(None, np.zeros(3), G, L0)   # one registration per context at a time
    assert ei.value.code == -5
    Tf, Td, t = g.register_wait()
    # hand-over: source := old target moved by a pure translation keeps the target's covariances
    shift = np.array([0.25, -0.1, 0.05], np.float32)
    moved = tgt.copy(); moved[:, :3] += shift
    ref = RotVGICP(); ref.setPolarResolution(*cfg["polar"]); ref.setInputTarget(tgt); ref.setInputSource(moved); ref.computeCovariances()
    g.setInputSource(moved); g.adoptTargetCovariances()
    assert np.abs(g.getSourceCovariances() - ref.getSourceCovariances()).max() < 1e-5   # equal up to the float rounding of the moved points
    with pytest.raises(RoloError) as ei:
        g.adoptTargetCovariances()             # already handed over
    assert ei.value.code == -5

    b = RotVGICPBatch(2)
    with pytest.raises(RoloError) as ei:
        b.register_async(None, None, np.tile(G, (2, 1)), np.tile(L0, (2, 1)))   # members without clouds
    assert ei.value.code == -5
    for m in b.members:
        m.setPolarResolution(*cfg["polar"]); m.setInputTarget(tgt); m.setInputSource(src)
    b.members[1].setFixedIterations(5)
    with pytest.raises(RoloError) as ei:
        b.register_async(None, None, np.tile(G, (2, 1)), np.tile(L0, (2, 1)))   # members must share the schedule
    assert ei.value.code == -7
    b.members[1].setFixedIterations(0)
    b.register_async(None, None, np.tile(G, (2, 1)), np.tile(L0, (2, 1)))
    with pytest.raises(RoloError) as ei:
        b.register_async(None, None, np.tile(G, (2, 1)), np.tile(L0, (2, 1)))
    assert ei.value.code == -5
    Tb, Tdb, tb = b.register_wait()
    assert np.abs(Tdb[0] - Td).max() < 1e-11 and np.abs(Tdb[1] - Td).max() < 1e-11 and np.abs(tb[0] - t).max() < 1e-11
    b.close()


def test_transform_cloud_matches_pcl_restatement():
    src, _, _ = make_pair("vlp16_polar")
    pts = np.zeros((src.shape[0], 8), np.float32); pts[:, :4] = src; pts[:, 3] = 1.0; pts[:, 4] = src[:, 3]
    T = np.eye(4, dtype=np.float32); T[:3, :3] = synth.rpy_to_R(0.01, 0.02, 0.03).astype(np.float32); T[:3, 3] = (0.3, -0.1, 0.05)
    g = RotVGICP()
    assert np.array_equal(g.transformPointCloud(pts, T), pyorc.transform_cloud_f(pts, T))


# ---- full-size, size-independent properties (BASELINE configs[1], the 128k headline frame, configs[2] = 262k) ----
@pytest.mark.parametrize("sensor,leaf", [("os1-64", 1.0), ("os1-128", 0.5), ("os1-128x2048", 0.5)])
def test_full_size_properties(sensor, leaf):
    src, tgt, (R, t) = synth.dense_pair(sensor)
    n = src.shape[0]
    assert n in (65536, 131072, 262144)
    g = RotVGICP(); g.setResolution(leaf); g.setFixedIterations(20)
    g.setInputTarget(tgt); g.setInputSource(src)
    # (0) north_star: "bit-exact voxel/hash indices" AT THE SIZES IT IS QUOTED ON (round 4's verdict: until then asserted at 14k / 16k points only) — every target
    # point's voxel key, the set of occupied voxels and their counts against the oracle (vmp_voxel.hpp:199-211)
    assert np.array_equal(g.targetVoxelKeys(), pyorc.voxel_keys(tgt, 1, leaf))
    p = pyorc.default_params(voxel_type=1, voxel_resolution=leaf, fixed_iterations=20)
    o = pyorc.Reg(p); o.set_target(tgt); o.set_source(src)
    # (1) voxel map conserves mass: counts sum to N_t, count-weighted mean of voxel means = cloud mean
    g.buildVoxelMap()
    k, c, m, v = g.voxels()
    assert o.build_voxelmap() == 0
    ko, co, _, _ = o.voxels()
    so_, sg_ = np.lexsort(ko.T[::-1]), np.lexsort(k.T[::-1])
    assert k.shape == ko.shape and np.array_equal(k[sg_], ko[so_]) and np.array_equal(c[sg_], co[so_])
    assert c.sum() == n and len(np.unique(k, axis=0)) == k.shape[0]
    assert np.abs((m[:, :3] * c[:, None]).sum(0) / n - tgt[:, :3].astype(np.float64).mean(0)).max() < 1e-9
    # (2) every PLANE covariance has singular values (1, 1, 1e-3)
    cov = g.getTargetCovariances()[::97, :3, :3]
    sv = np.linalg.svd(cov, compute_uv=False)
    assert np.abs(sv - np.array([1, 1, 1e-3])).max() < 1e-9
    # (3) linearity of the reduction: H, b, err over a split of the source = sum of the parts (same target map)
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.003, -0.002, 0.01)
    e, H, b = g.so3_linearize(T)
    # (3a) the correspondence list of that linearisation, bit-exact at full size: the same source points find a voxel, and the same voxel (rot_vgicp_impl.hpp:173-200)
    eo, Ho, bo = o.so3_linearize(T)
    s_o, v_o = o.correspondences()
    found, keys = g.correspondences()
    assert np.array_equal(np.nonzero(found[:, 0])[0], np.sort(s_o))
    order = np.argsort(s_o, kind="stable")
    assert np.array_equal(keys[s_o[order], 0], ko[v_o[order]])
    assert abs(e - eo) <= 1e-9 * abs(eo) and np.abs(H - Ho).max() <= 1e-9 * np.abs(Ho).max() and np.abs(b - bo).max() <= 1e-9 * np.abs(bo).max()
    parts = []
    for sl in (slice(0, n // 2), slice(n // 2, n)):
        gp = RotVGICP(); gp.setResolution(leaf)
        gp.setInputTarget(tgt); gp.setInputSource(src[sl])
        gp.computeCovariances()
        gp.setSourceCovariances(g.getSourceCovariances()[sl])  # neighbourhoods of the full cloud
        parts.append(gp.so3_linearize(T))
    assert abs(parts[0][0] + parts[1][0] - e) <= 1e-11 * abs(e)
    assert np.abs(parts[0][1] + parts[1][1] - H).max() <= 1e-11 * np.abs(H).max()
    assert np.abs(parts[0][2] + parts[1][2] - b).max() <= 1e-11 * np.abs(b).max()
    # (4) the solve runs exactly 20 outer iterations, returns a rotation, and re-running reproduces it
    Tf = g.align(); T1 = g.final_transformation_d.copy()
    assert g.last_stats.n_outer == 20
    assert np.abs(T1[:3, :3] @ T1[:3, :3].T - np.eye(3)).max() < 1e-12 and np.all(T1[:3, 3] == 0)
    g.align()
    assert np.abs(g.final_transformation_d - T1).max() < 1e-10   # (the voxel sums are order-independent integers; the tolerance covers the different row shapes of the passes)
    # (5) and agrees with the oracle end to end at full size
    rc, _, Td_o, it_o, _ = o.align()
    assert rc == 0 and rot_angle(T1[:3, :3], Td_o[:3, :3]) <= 1e-5
    t_g = g.computeTranslation(np.zeros(3), G, L0)
    rc, t_o, _ = o.compute_translation(np.zeros(3), G, L0)
    assert np.abs(t_g - t_o).max() <= 1e-4


@pytest.mark.parametrize("sensor,n_scan,horizon", [("vlp16", 16, 1800), ("os1-128", 128, 1024)])
def test_full_size_polar_pipeline_frame_indices_bit_exact(sensor, n_scan, horizon):
    """The drop-in pipeline's registration inputs at their own size: the feature clouds of two consecutive raw frames (reference thinning), POLAR voxels
    0.175 / 0.175 / 2.0 as lidarOdometry.cpp:462 sets them — every target point's voxel key, the occupied voxels and counts, and the correspondence list at
    a probe rotation, all bit-exact against the oracle (vmp_voxel.hpp:208-211 through atan2 / acos; rot_vgicp_impl.hpp:173-200)."""
    cfg = dict(n_scan=n_scan, horizon_scan=horizon)
    fo = pyorc.front_params(**cfg)
    clouds = []
    for k_, (R_, t_) in enumerate([(np.eye(3), np.zeros(3)), (synth.rpy_to_R(*np.deg2rad([0.5, 1.0, 2.0])), np.array([0.30, 0.05, 0.02]))]):
        fr = synth.make_frame(sensor, R_, t_, synth.SEED + k_)
        ex = pyorc.extract_features(fo, pyorc.project(fo, fr.xyz, fr.ring))
        clouds.append(np.ascontiguousarray(np.concatenate([ex["corner"], ex["surface"]])[:, :4], np.float32))
    src, tgt = clouds
    polar = (0.175, 0.175, 2.0)
    p = pyorc.default_params(polar_resolution=polar)
    o = pyorc.Reg(p); o.set_target(tgt); o.set_source(src)
    g = RotVGICP(); g.setPolarResolution(*polar); g.setInputTarget(tgt); g.setInputSource(src)
    assert np.array_equal(g.targetVoxelKeys(), pyorc.voxel_keys(tgt, 0, 1.0, polar))
    g.buildVoxelMap(); assert o.build_voxelmap() == 0
    kg, cg, _, _ = g.voxels(); ko, co, _, _ = o.voxels()
    so_, sg_ = np.lexsort(ko.T[::-1]), np.lexsort(kg.T[::-1])
    assert kg.shape == ko.shape and np.array_equal(kg[sg_], ko[so_]) and np.array_equal(cg[sg_], co[so_])
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.004, -0.007, 0.02)
    eo, Ho, bo = o.so3_linearize(T); eg, Hg, bg = g.so3_linearize(T)
    s_o, v_o = o.correspondences()
    found, keys = g.correspondences()
    assert len(s_o) > 0.5 * src.shape[0]
    assert np.array_equal(np.nonzero(found[:, 0])[0], np.sort(s_o))
    order = np.argsort(s_o, kind="stable")
    assert np.array_equal(keys[s_o[order], 0], ko[v_o[order]])
    assert abs(eg - eo) <= 1e-9 * abs(eo) and np.abs(Hg - Ho).max() <= 1e-9 * np.abs(Ho).max()


def test_point_sharded_passes_sum_to_the_full_result():
    """SURVEY §8e on one GPU: contexts sharded as rank r of W (test hook, no communicator) return partial sums whose
    total equals the unsharded pass — what the RCCL all-reduce of the 32 fp64 computes on W GPUs."""
    import ctypes as C
    from rolo_amd._lib import lib, check
    src, tgt, cfg = make_pair("os64_uniform")
    _, g = make_both(src, tgt, cfg)
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.004, -0.007, 0.02)
    e, H, b = g.so3_linearize(T)
    tp = np.array([0.01, -0.004, 0.002])
    et, Ht, bt = g.t3_linearize(tp, G, L0)
    nfound = int(g.correspondences()[0].sum())
    for W in (2, 4, 8):
        acc = [0.0, np.zeros((3, 3)), np.zeros(3), 0.0, np.zeros((6, 6)), np.zeros(6), 0]
        for r in range(W):
            _, gr = make_both(src, tgt, cfg)
            check(lib().rolo_set_shard(gr._h, r, W), "rolo_set_shard")
            er, Hr, br = gr.so3_linearize(T)
            acc[0] += er; acc[1] += Hr; acc[2] += br
            acc[6] += int(gr.correspondences()[0].sum())
            # NB: lambda_/pt_size uses the GLOBAL correspondence count in the real multi-GPU path (it is all-reduced);
            # with the test hook each shard only knows its own count, so compare the CT-free part: ct_lambda = 0
            e0, H0, b0 = gr.t3_linearize(tp, G, L0, ct_lambda=0.0)
            acc[3] += e0; acc[4] += H0; acc[5] += b0
        assert acc[6] == nfound
        assert abs(acc[0] - e) <= 1e-12 * abs(e)
        assert np.abs(acc[1] - H).max() <= 1e-12 * np.abs(H).max() and np.abs(acc[2] - b).max() <= 1e-12 * np.abs(b).max()
        e0f, H0f, b0f = g.t3_linearize(tp, G, L0, ct_lambda=0.0)
        assert abs(acc[3] - e0f) <= 1e-12 * abs(e0f) and np.abs(acc[4] - H0f).max() <= 1e-12 * np.abs(H0f).max()
        g.so3_linearize(T)  # restore the cached correspondences of the probe pose


def test_cpp_shim_end_to_end(tmp_path):
    """fast_gicp::RotVGICP from include/rot_vgicp_hip.hpp, driven as lidarOdometry.cpp:460-500 does, in a C++-only
    process (no Python / torch): same result as the oracle."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "shim_demo")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "shim_demo.cpp"), "-o", exe,
           "-L", os.path.join(root, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(root, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    src, tgt, cfg = make_pair("vlp16_polar")
    src.astype(np.float32).tofile(tmp_path / "s.bin"); tgt.astype(np.float32).tofile(tmp_path / "t.bin")
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "t.bin")], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().splitlines()
    vals = [float(v) for v in lines[0].split()]
    T = np.array(vals[:16]).reshape(4, 4); t = np.array(vals[16:19])
    o, _ = make_both(src, tgt, cfg)
    rc, Tf_o, Td_o, it, cv = o.align()
    g3 = np.array([-0.28, -0.04, -0.02])
    rc2, t_o, _ = o.compute_translation(np.zeros(3), g3, g3)
    assert np.abs(T - Tf_o).max() < 1e-6 and np.abs(t - t_o).max() <= 1e-4
    assert int(vals[19]) == int(cv) and int(vals[20]) == src.shape[0]
    assert int(vals[21]) == 1   # `aligned` (host-side pcl::transformPointCloud of the class) = rolo_transform_cloud, bit for bit
    # setSource/TargetCovariances, getSource/TargetCovariances, evaluateCost, getFinalHessian (rot_vgicp.hpp:89-97, lsq_registration.hpp:55-57):
    # an operator fed with the first one's covariances reproduces its rotation; evaluateCost is the oracle's 6-dof linearisation
    v2 = [float(v) for v in lines[1].split()]
    assert v2[0] < 1e-6 and int(v2[5]) == src.shape[0]
    e_o, H_o, b_o = o.linearize(np.eye(4))
    assert abs(v2[1] - e_o) <= 1e-9 * abs(e_o) and abs(v2[2] - np.trace(H_o)) <= 1e-9 * abs(np.trace(H_o)) and abs(v2[3] - np.linalg.norm(b_o)) <= 1e-9 * np.linalg.norm(b_o)
    assert v2[4] == 1.0   # final_hessian_ stays Identity on the SO(3) optimiser (lsq_registration_impl.hpp:16-24)
    assert lines[2] == "invalid_argument"


def test_cpp_shim_debug_print_is_the_references_table(tmp_path):
    """setDebugPrint(true) (lsq_registration.hpp:60): the class prints computeTransformation's banner and, per outer iteration, the header and one row per LM
    trial in the reference's boost::format layout (lsq_registration_impl.hpp:158-162, :299-305 for the rotation stage, :114-120 for the translation stage) —
    here from the device-side trace; the rows must be the trace of the same solve, formatted with the same conversions"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "shim_demo")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "shim_demo.cpp"), "-o", exe,
           "-L", os.path.join(root, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(root, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    src, tgt, cfg = make_pair("vlp16_polar")
    src.astype(np.float32).tofile(tmp_path / "s.bin"); tgt.astype(np.float32).tofile(tmp_path / "t.bin")
    r = subprocess.run([exe, "debug", str(tmp_path / "s.bin"), str(tmp_path / "t.bin")], capture_output=True, text=True, check=True)
    rot_txt, trans_txt = r.stdout.split("=== translation ===\n")
    g = RotVGICP(); g.setPolarResolution(*cfg["polar"]); g.setInputTarget(tgt); g.setInputSource(src)
    g.align(); g3 = np.array([-0.28, -0.04, -0.02]); g.computeTranslation(np.zeros(3), g3, g3)
    tr = g.trace()
    head = "--- LM optimization ---\n%5s %15s %15s %15s %15s %15s %5s\n" % ("i", "y0", "yi", "rho", "lambda", "|delta|", "dec")

    def table(stage):
        out = ""
        for rec in tr:
            if rec["stage"] != stage:
                continue
            if rec["trial"] == 0:
                out += head
            out += "%5d %15g %15g %15g %15g %15g %5s\n" % (rec["trial"], rec["y0"], rec["yi"], rec["rho"], rec["lam"], rec["dnorm"], "x" if rec["rho"] > 0 else " ")
        return out
    banner = "*" * 44 + "\n" + "*" * 17 + " optimize " + "*" * 17 + "\n" + "*" * 44 + "\n"
    assert rot_txt == banner + table(0)
    assert trans_txt == table(1)
    assert rot_txt.count("--- LM optimization ---") >= 2 and trans_txt.count("--- LM optimization ---") >= 1


def test_cpp_operator_constructed_per_frame_uses_the_context_pool(tmp_path):
    """src/lidarOdometry.cpp:460 constructs `fast_gicp::RotVGICP rot_vgicp;` inside scanRegeistration, once per frame. The drop-in class
    does that through rolo_ctx_acquire / _release: 50 frames with the object constructed inside the frame give the persistent object's
    results bit for bit, at (about) its latency, while an unpooled context per frame would add a multiple of the whole frame."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "shim_demo")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "shim_demo.cpp"), "-o", exe,
           "-L", os.path.join(root, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(root, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    src, tgt, cfg = make_pair("vlp16_polar")
    src.astype(np.float32).tofile(tmp_path / "s.bin"); tgt.astype(np.float32).tofile(tmp_path / "t.bin")
    r = subprocess.run([exe, "loop", str(tmp_path / "s.bin"), str(tmp_path / "t.bin"), "50"], capture_output=True, text=True, check=True)
    per_frame_obj, persistent, create_destroy, same_a, same_b = r.stdout.split()
    print("ms per frame: object per frame %s, persistent object %s; an unpooled rolo_ctx_create + destroy alone: %s" % (per_frame_obj, persistent, create_destroy))
    assert same_a == "1" and same_b == "1"                      # every frame of both loops reproduces the first frame's result exactly
    # the measured figure (bench.py `cpp_operator_per_frame`) is not this test's business — a shared box can stretch either loop; what it must catch
    # is an operator whose construction costs a multiple of the frame. (Until round 4 an unpooled rolo_ctx_create + destroy cost 8-10 ms, mostly its two
    # hipStreamCreate; with the per-device stream bank it is ~0.7 ms — still a whole frame, which the pool saves.)
    assert float(per_frame_obj) <= 2.0 * float(persistent) + 0.5


def test_context_pool_hands_out_fresh_objects():
    """a released context comes back with the defaults of a new one: parameters, no clouds, no cached results"""
    import ctypes as C
    from rolo_amd._lib import lib, Params
    src, tgt, cfg = make_pair("vlp16_polar")
    L = lib()
    L.rolo_ctx_pool_clear()
    h = C.c_void_p(); assert L.rolo_ctx_acquire(0, C.byref(h)) == 0
    g = RotVGICP.__new__(RotVGICP); RotVGICP.__init__(g)   # an ordinary context next to it
    first = h.value
    p = Params(); L.rolo_default_params(C.byref(p)); p.k_correspondences = 10; p.voxel_type = 1; p.voxel_resolution = 0.7
    assert L.rolo_set_params(h, C.byref(p)) == 0
    a = np.ascontiguousarray(tgt, np.float32)
    assert L.rolo_set_target(h, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1]) == 0
    assert L.rolo_set_source(h, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1]) == 0
    assert L.rolo_compute_covariances(h) == 0 and L.rolo_build_voxelmap(h) == 0 and L.rolo_num_voxels(h) > 0
    L.rolo_ctx_release(h)
    h2 = C.c_void_p(); assert L.rolo_ctx_acquire(0, C.byref(h2)) == 0
    assert h2.value == first                                     # the parked context, not a new one
    assert L.rolo_num_voxels(h2) == -5                           # ROLO_ESTATE: no map
    assert L.rolo_compute_covariances(h2) == -5                  # no clouds
    cov = np.zeros((4, 16)); assert L.rolo_get_source_covariances(h2, cov.ctypes.data_as(C.POINTER(C.c_double))) == -5
    # default parameters again: k = 20 covariances on a fresh pair equal a brand-new context's
    assert L.rolo_set_target(h2, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1]) == 0
    assert L.rolo_set_source(h2, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[0], a.shape[1]) == 0
    assert L.rolo_compute_covariances(h2) == 0
    c2 = np.zeros((a.shape[0], 16)); assert L.rolo_get_target_covariances(h2, c2.ctypes.data_as(C.POINTER(C.c_double))) == 0
    g.setInputTarget(tgt); g.setInputSource(tgt.copy()); g.computeCovariances()
    assert np.array_equal(c2.reshape(-1, 4, 4), g.getTargetCovariances())
    L.rolo_ctx_release(h2); L.rolo_ctx_pool_clear(); g.close()


@pytest.mark.parametrize("guess_deg", [0.0, 4.0])
def test_fused_lm_equals_pass_plus_controller_launches(guess_deg):
    """rolo_params.fused_lm (one launch per LM trial) against the pass + controller launches on every driver that can run it: rolo_align +
    rolo_compute_translation (synchronous chunks of 8 — an EVEN chunk, whose closing launch reads and writes the same state buffer — and, with
    a guess 4 degrees off, several top-up chunks), and rolo_register_async (odd first schedule, top-ups in rolo_register_wait). Same trial
    sequence, poses equal to rounding (the two forms sum the per-workgroup rows in different shapes)."""
    src, tgt, cfg = make_pair("os64_uniform")
    guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.0, 0.0, np.deg2rad(guess_deg))
    out = []
    for fused in (0, 1, 2):   # 2: one launch per FRAME (the resident LM kernel, round 6)
        g = RotVGICP(); g.setResolution(cfg["leaf"]); g.setFusedLm(fused); g.setUseGraph(False)
        if guess_deg > 0:
            g.setRotationEpsilon(1e-9); g.setInitialLambdaFactor(100.0)   # heavily damped steps, iterated long after the reference's 2e-3 would stop: more passes than the first chunk of 8 holds
        g.setInputTarget(tgt); g.setInputSource(src)
        g.align(guess)
        Td = g.final_transformation_d.copy(); st = (g.last_stats.n_outer, g.last_stats.n_passes, g.last_stats.converged)
        t = g.computeTranslation(np.zeros(3), G, L0)
        tr = g.trace()
        g2 = RotVGICP(); g2.setResolution(cfg["leaf"]); g2.setFusedLm(fused); g2.setUseGraph(bool(fused == 2))   # (the resident kernel also through a captured graph: frame 3 replays)
        if guess_deg > 0:
            g2.setRotationEpsilon(1e-9); g2.setInitialLambdaFactor(100.0)
        res = []
        for k in range(3):   # frame 1: first schedule from the defaults; later frames: from the hints (odd / even lengths both occur)
            g2.setInputTarget(tgt.copy()); g2.setInputSource(src.copy())
            g2.register_async(guess, np.zeros(3), G, L0)
            Tf2, Td2, t2 = g2.register_wait()
            res.append((Td2.copy(), t2.copy(), g2.last_stats.n_passes, g2.last_translation_stats.n_passes))
        out.append((Td, st, t, tr, res, g2.counters()))
        g.close(); g2.close()
    (Ta, sa, ta, tra, ra, ca) = out[0]
    for (Tb, sb, tb, trb, rb, cb) in out[1:]:
        assert sa == sb and len(tra) == len(trb)
        assert [(r["stage"], r["outer"], r["trial"], r["accepted"]) for r in tra] == [(r["stage"], r["outer"], r["trial"], r["accepted"]) for r in trb]
        assert np.abs(Ta - Tb).max() < 1e-11 and np.abs(ta - tb).max() < 1e-11
        for (Tda, t_a, pa, qa), (Tdb, t_b, pb, qb) in zip(ra, rb):
            assert (pa, qa) == (pb, qb) and np.abs(Tda - Tdb).max() < 1e-11 and np.abs(t_a - t_b).max() < 1e-11
            assert np.abs(Tda - Ta).max() < 1e-11   # and the asynchronous driver agrees with the synchronous ones
    assert out[2][5]["topup_frames"] == 0   # the resident kernel runs until the state says the frame is over: no schedule to fall short of
    if guess_deg > 0:
        assert sa[1] > 8   # the rotation stage did need more than the first chunk: the top-up path ran


_SPEC_AB = r"""
import sys, hashlib, json, numpy as np
sys.path.insert(0, sys.argv[1])
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP
G = -np.asarray(synth.PREV_STEP_T); L0 = G * 0.97
out = {}
import bench
for name, kind, stride, leaf, polar, fixed, lm_init in (("uniform20", "os1-64", 4, 1.0, None, 20, None), ("polar", "vlp16", 2, None, (0.175, 0.175, 2.0), 0, None),
                                                        ("pool7", "os1-128", 1, 0.5, None, 20, None)):
    # pool7: pair 7 of bench.py's input pool — its translation stage ends "... R R A A": acceptances right after rejections
    src, tgt = bench._pool_pair((kind, synth.SEED, 7)) if name == "pool7" else synth.dense_pair(kind, col_stride=stride)[:2]
    for fused in (0, 1):
        g = RotVGICP(); g.setUseGraph(False); g.setFusedLm(bool(fused))
        g.setResolution(leaf) if leaf else g.setPolarResolution(*polar)
        g.setFixedIterations(fixed)
        if lm_init: g.setInitialLambdaFactor(lm_init); g.setRotationEpsilon(1e-7)
        g.setInputTarget(tgt); g.setInputSource(src)
        g.register_async(None, np.zeros(3), G, L0); Tf, Td, t = g.register_wait()
        tr = g.trace()
        out[name + str(fused)] = dict(T=hashlib.sha256(Td.tobytes()).hexdigest(), t=hashlib.sha256(np.asarray(t).tobytes()).hexdigest(), outer=(g.last_stats.n_outer, g.last_translation_stats.n_outer),
                         passes=(g.last_stats.n_passes, g.last_translation_stats.n_passes), pattern="".join("RAC"[r["accepted"]] for r in tr),
                         trace=hashlib.sha256(json.dumps([[r[k] for k in ("stage", "outer", "trial", "accepted", "y0", "yi", "rho", "lam", "dnorm")] for r in tr]).encode()).hexdigest())
        g.close()
print(json.dumps(out))
"""


def test_speculative_linearisation_skipping_changes_no_bit():
    """LmState::lin_skip (round 5): after a rejected trial the next pass evaluates its trial's cost alone; if that trial is accepted after all, one more pass
    linearises at the accepted pose. Against ROLO_LM_SPEC_LIN=0 (every pass carries both halves, rounds 1-4): poses, translations, iteration counts and the WHOLE LM
    trace are the same bits — 20 forced iterations (a run of rejected-but-converged trials), a convergence-driven POLAR solve, and a frame of the bench's pool whose
    translation stage accepts right after rejections (the mis-predicted case with its extra linearise-only pass), each with pass + controller launches and with one
    launch per trial."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for spec in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _SPEC_AB, root], env=dict(os.environ, ROLO_LM_SPEC_LIN=spec), capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    on, off = res
    for k in on:
        for f in ("T", "t", "outer", "pattern", "trace"):
            assert on[k][f] == off[k][f], (k, f, on[k], off[k])
    assert "C" in on["uniform200"]["pattern"] or "R" in on["uniform200"]["pattern"]           # there were rejected trials to skip after
    assert any("RA" in on[k]["pattern"] or "CA" in on[k]["pattern"] for k in on), {k: on[k]["pattern"] for k in on}   # ... and an acceptance right after a rejection (the extra pass)
    assert any(on[k]["passes"] != off[k]["passes"] for k in on)                                    # which shows as extra linearise-only passes, nothing else


def test_load_hint_picks_the_walk_and_the_result_does_not_depend_on_it():
    """rolo_set_load_hint / the per-frame choice of round 5: a large launch takes the 64-query packet walk (rolo_ctx_counters "walk_lanes" = 1) when the device is
    busy and two lanes per query when it is idle; pinned by the hint, decided from the frames other contexts have in flight otherwise (sticky for a few frames).
    Both walks return the same lists, so the covariances are the same BITS. Since round 6 the hint also sizes the resident LM kernel (256 workgroups on an idle device,
    64 beside other frames): the rows of a pass are then summed over other groups of points, and poses agree to rounding (1e-12) instead of bit for bit — with
    fused_lm = 0 they still are the same bits."""
    src, tgt, _ = synth.dense_pair("os1-128")
    assert src.shape[0] == 131072
    z = np.zeros(3)

    def ctx(hint, fused=2):
        g = RotVGICP(); g.setResolution(0.5); g.setFixedIterations(20); g.setUseGraph(False); g.setLoadHint(hint); g.setFusedLm(fused)
        g.setInputTarget(tgt); g.setInputSource(src)
        return g
    res = {}
    for fused in (0, 2):
        for hint in (0, 1):
            g = ctx(hint, fused)
            g.register_async(None, z, G, L0); Tf, Td, t = g.register_wait()
            res[fused, hint] = (Td.copy(), np.asarray(t).copy(), g.counters()["walk_lanes"], g.getSourceCovariances().copy()); g.close()
        assert res[fused, 0][2] == 2 and res[fused, 1][2] == 1
        assert np.array_equal(res[fused, 0][3], res[fused, 1][3])                      # the search's results do not depend on the walk
    assert np.array_equal(res[0, 0][0], res[0, 1][0]) and np.array_equal(res[0, 0][1], res[0, 1][1])   # pass + controller launches: the same bits
    for a_, b_ in ((res[2, 0], res[2, 1]), (res[2, 0], res[0, 0])):
        assert np.abs(a_[0] - b_[0]).max() < 1e-12 and np.abs(a_[1] - b_[1]).max() < 1e-12
    res = {0: res[2, 0], 1: res[2, 1]}
    # automatic: a alone -> the idle-device walk; b enqueued while a's frame is in flight -> the busy-device walk; b stays with it for its next frames (no flip per caller barrier)
    a, b = ctx(-1), ctx(-1)
    a.register_async(None, z, G, L0)
    b.register_async(None, z, G, L0)
    Ta = a.register_wait()[1]; Tb = b.register_wait()[1]
    assert a.counters()["walk_lanes"] == 2 and b.counters()["walk_lanes"] == 1
    assert np.array_equal(Ta, res[0][0]) and np.array_equal(Tb, res[1][0])   # a: the idle-device sizes, b: the busy-device ones
    b.setInputTarget(tgt); b.setInputSource(src)
    b.register_async(None, z, G, L0); b.register_wait()
    assert b.counters()["walk_lanes"] == 1
    import ctypes as C
    from rolo_amd._lib import lib
    assert lib().rolo_set_load_hint(b._h, 2) < 0 and lib().rolo_set_load_hint(None, 0) < 0
    a.close(); b.close()


def test_non_finite_covariance_in_the_voxel_build_is_an_error():
    """a NaN covariance (degenerate neighbourhood) must not become a finite but wrong voxel: ROLO_ENONFINITE (-11)"""
    from rolo_amd._lib import RoloError
    src, tgt, cfg = make_pair("vlp16_polar")
    g = RotVGICP(); g.setPolarResolution(0.175, 0.175, 2.0)
    g.setInputTarget(tgt); g.setInputSource(src)
    g.computeCovariances()
    ct = g.getTargetCovariances().copy(); ct[7, :3, :3] = np.nan
    g.setTargetCovariances(ct)     # handed-in covariances go through fp64 atomics: the NaN propagates as in the reference (no error)
    g.buildVoxelMap()
    g2 = RotVGICP(); g2.setPolarResolution(0.175, 0.175, 2.0)
    bad = tgt.copy(); bad[5, 0] = np.inf
    g2.setInputTarget(bad); g2.setInputSource(src)
    with pytest.raises(RoloError) as ei:
        g2.register_async(None, np.zeros(3), G, L0); g2.register_wait()
    assert ei.value.code in (-11, -10)   # the non-finite point is caught by the fixed-point sums or, first, by the key range check
    g.close(); g2.close()


def test_config5_distinct_full_size_pairs_through_four_contexts_with_graph_replay():
    """BASELINE configs[4] in its one-GPU form, under pytest: 32 DISTINCT full-size OS1-64 pairs (65 536 points per cloud, seeds 20260926 + i,
    motions U(+-2 deg), U(+-0.4 m)), UNIFORM leaf 1.0 m, 20 SO(3) LM iterations + CT translation, resident in HBM, streamed through four
    contexts that replay the hipGraph of their frame — EVERY pose against the oracle (bar: 1e-5 rad / 1e-4 m)."""
    import multiprocessing as mp
    import torch
    import bench
    npairs = 32
    with mp.get_context("spawn").Pool(min(bench.usable_cores(), 16)) as pool:
        pairs = pool.map(bench._config5_pair, list(range(npairs)), chunksize=2)
    dev = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda(), s.shape[0], t.shape[0]) for s, t in pairs]
    assert all(d[2] == 65536 and d[3] == 65536 for d in dev)
    ctxs = []
    for _ in range(4):
        g = RotVGICP(); g.setResolution(1.0); g.setFixedIterations(20); ctxs.append(g)
    zero3 = np.zeros(3)
    results = [None] * npairs

    def enqueue(g, k):
        s, t, ns, nt = dev[k]
        g.setInputTargetDevice(t.data_ptr(), nt, 4); g.setInputSourceDevice(s.data_ptr(), ns, 4)
        g.register_async(None, zero3, G, L0, 0.1, 0.1, 0.3)
    for sweep in range(2):   # the second sweep runs on replayed graphs throughout
        nxt = 0; inflight = []
        for g in ctxs:
            enqueue(g, nxt); inflight.append((g, nxt)); nxt += 1
        while inflight:
            g, k = inflight.pop(0)
            Tf, Td, t = g.register_wait()
            results[k] = (Td.copy(), t.copy())
            if nxt < npairs:
                enqueue(g, nxt); inflight.append((g, nxt)); nxt += 1
    cnt = [g.counters() for g in ctxs]
    assert sum(c["graph_replays"] for c in cnt) >= npairs   # the graphs were replayed, not re-captured frame after frame
    p = pyorc.default_params(voxel_type=pyorc.VOXEL_UNIFORM, voxel_resolution=1.0, fixed_iterations=20, num_threads=bench.usable_cores())
    worst_r = worst_t = 0.0
    for k in range(npairs):
        o = pyorc.Reg(p); o.set_target(pairs[k][1]); o.set_source(pairs[k][0])
        rc, _, Td, _, _ = o.align(); rc2, to, _ = o.compute_translation(np.zeros(3), G, L0)
        assert rc == 0 and rc2 == 0
        Tg, tg = results[k]
        dR = Tg[:3, :3] @ Td[:3, :3].T
        worst_r = max(worst_r, float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))); worst_t = max(worst_t, float(np.abs(tg - to).max()))
    assert worst_r <= 1e-5 and worst_t <= 1e-4, (worst_r, worst_t)
    for g in ctxs:
        g.close()


def test_graph_replay_equals_eager():
    """rolo_register_async captures the frame's launch schedule in a hipGraph on the second identical-shape frame and
    replays it afterwards; per-frame arguments (guess, translations) are refreshed through a captured H2D copy."""
    import torch
    src, tgt, cfg = make_pair("os64_uniform")
    d_src = torch.from_numpy(src).cuda(); d_tgt = torch.from_numpy(tgt).cuda()
    n = src.shape[0]

    def run(use_graph, frames):
        g = RotVGICP(); g.setResolution(cfg["leaf"]); g.setFixedIterations(20); g.setUseGraph(use_graph)
        out = []
        for k in range(frames):
            g.setInputTargetDevice(d_tgt.data_ptr(), n, 4); g.setInputSourceDevice(d_src.data_ptr(), n, 4)
            guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.001 * k, 0.0, 0.002 * k)
            g.register_async(guess, np.array([0.001 * k, 0, 0]), G * (1 + 0.01 * k), L0)
            Tf, Td, t = g.register_wait()
            out.append((Td.copy(), t.copy(), g.last_stats.n_outer, g.last_translation_stats.n_outer))
        return out

    eager = run(False, 5)
    graph = run(True, 5)   # frame 0 eager, frame 1 captured + launched, frames 2-4 replayed
    for (Te, te, ie, je), (Tg, tg, ig, jg) in zip(eager, graph):
        assert np.abs(Tg - Te).max() < 1e-11 and np.abs(tg - te).max() < 1e-11
        assert (ie, je) == (ig, jg)
    assert np.abs(eager[0][1] - eager[3][1]).max() > 1e-6  # the frames really differ (arguments were refreshed)


def test_batch_equals_individual():
    """rolo_batch_*: B scan pairs of different sizes registered as one batch (shared pass / controller launches, replayed
    from a hipGraph) give what B separate register_async/wait calls give — including a member that runs the normal LM
    schedule to convergence next to members with different data."""
    import torch
    from rolo_amd.rotvgicp import RotVGICPBatch
    pairs = [make_pair("os64_uniform"), make_pair("vlp16_polar"), make_pair("os64_uniform")]
    dev = [(torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()) for s, t, _ in pairs]

    def configure(g, i):
        cfg = pairs[i][2]
        if cfg["voxel_type"] == 0:
            g.setPolarResolution(*cfg["polar"])
        else:
            g.setResolution(cfg["leaf"])
        g.setInputTargetDevice(dev[i][1].data_ptr(), pairs[i][1].shape[0], 4)
        g.setInputSourceDevice(dev[i][0].data_ptr(), pairs[i][0].shape[0], 4)

    def guess(i, k):
        T = np.eye(4, dtype=np.float32); T[:3, :3] = synth.rpy_to_R(0.001 * k, 0.0005 * i, 0.002 * k)
        return T

    frames = 4
    single = []
    for i in range(3):
        g = RotVGICP(); rows = []
        for k in range(frames):
            configure(g, i)
            g.register_async(guess(i, k), np.array([0.001 * k, 0.0, 0.001 * i]), G * (1 + 0.01 * k), L0)
            Tf, Td, t = g.register_wait()
            rows.append((Td.copy(), t.copy(), g.last_stats.n_outer, g.last_translation_stats.n_outer, g.last_stats.n_correspondences))
        single.append(rows)

    b = RotVGICPBatch(3)
    for k in range(frames):   # frame 0 eager, 1 captured, 2.. replayed
        for i, m in enumerate(b.members):
            configure(m, i)
        b.register_async(np.stack([guess(i, k) for i in range(3)]), np.array([[0.001 * k, 0.0, 0.001 * i] for i in range(3)]),
                         np.tile(G * (1 + 0.01 * k), (3, 1)), np.tile(L0, (3, 1)))
        Tf, Td, t = b.register_wait()
        for i, m in enumerate(b.members):
            Ts, ts, io, jo, nc = single[i][k]
            assert np.abs(Td[i] - Ts).max() < 1e-11 and np.abs(t[i] - ts).max() < 1e-11, (k, i)
            assert (m.last_stats.n_outer, m.last_translation_stats.n_outer, m.last_stats.n_correspondences) == (io, jo, nc)
            assert np.abs(Tf[i] - Ts.astype(np.float32)).max() < 1e-6
    # a member of a batch is still a normal operator afterwards (stage-by-stage API on its own stream)
    err, H, bvec = b.members[1].so3_linearize(single[1][-1][0])
    assert np.isfinite(err) and np.isfinite(H).all()
    b.close()


def test_rccl_path_on_one_rank():
    """The point-sharded schedule — partial rows reduced on the device, ncclAllReduce of the 32 fp64 sums on the context's
    stream, controller on the reduced sums — driven on real hardware with a one-rank RCCL communicator (the pool has
    single-GPU boxes): same poses as the plain path. The multi-rank arithmetic is covered by the shard-linearity test above
    and the 2-rank gloo test."""
    src, tgt, cfg = make_pair("os64_uniform")
    ref = RotVGICP(); ref.setResolution(cfg["leaf"]); ref.setInputTarget(tgt); ref.setInputSource(src)
    ref.register_async(None, np.zeros(3), G, L0); Tf0, Td0, t0 = ref.register_wait()
    g = RotVGICP(); g.setResolution(cfg["leaf"])
    g.comm_init(RotVGICP.comm_unique_id(), 0, 1)
    for _ in range(3):   # eager frames: a context with a communicator does not capture graphs
        g.setInputTarget(tgt); g.setInputSource(src)
        g.register_async(None, np.zeros(3), G, L0)
        Tf, Td, t = g.register_wait()
        assert np.abs(Td - Td0).max() < 1e-11 and np.abs(t - t0).max() < 1e-11
        assert g.last_stats.n_outer == ref.last_stats.n_outer and g.last_translation_stats.n_outer == ref.last_translation_stats.n_outer
    # stage-level calls go through the collective as well
    T = np.eye(4); T[:3, :3] = synth.rpy_to_R(0.003, -0.002, 0.01)
    e1, H1, b1 = g.so3_linearize(T); e0, H0, b0 = ref.so3_linearize(T)
    assert abs(e1 - e0) <= 1e-11 * abs(e0) and np.abs(H1 - H0).max() <= 1e-11 * np.abs(H0).max()


@pytest.mark.parametrize("optimizer", [1, 0])   # 6-dof LevenbergMarquardt, GaussNewton
def test_ill_conditioned_normal_equations_match_the_pivoted_solve(optimizer):
    """The device controller solves (H + lambda I) d = -b with an UNPIVOTED LDL^T in registers; the reference (and the oracle) use Eigen's
    pivoted LDLT. A single noisy plane barely constrains the in-plane motion: the 6 x 6 H of the 6-dof optimisers has a condition number
    above 1e5 (plane-normal against in-plane weights, metres against radians), the solve takes 22 outer iterations — and the two
    factorisations must still walk the same path: same number of iterations, same accept / reject decisions, same pose to far below the bars."""
    g0 = np.random.default_rng(7)
    n = 6000
    xy = g0.uniform(-20, 20, (n, 2))
    tgt = np.c_[xy, 0.02 * g0.standard_normal(n), np.ones(n)].astype(np.float32)
    R = synth.rpy_to_R(0.004, -0.006, 0.01)
    src = tgt.copy(); src[:, :3] = (tgt[:, :3].astype(np.float64) @ R.T).astype(np.float32) + 0.01 * g0.standard_normal((n, 3)).astype(np.float32)
    src[:, :3] += np.float32([0.02, -0.01, 0.03])
    cfg = dict(voxel_type=1, polar=(0.175, 0.175, 2.0), leaf=2.0)
    o, g = make_both(src, tgt, cfg, optimizer=optimizer)
    rc, Tf_o, Td_o, it_o, cv_o = o.align()
    g.align(); Td_g = g.final_transformation_d
    e, H, b = o.linearize(np.eye(4))
    assert np.linalg.cond(H) > 1e5
    assert rc == 0 and g.last_stats.n_outer == it_o and it_o > 10
    assert rot_angle(Td_g[:3, :3], Td_o[:3, :3]) < 1e-8 and np.abs(Td_g[:3, 3] - Td_o[:3, 3]).max() < 1e-7
    tg, to = g.trace(), o.trace()
    assert len(tg) == len(to) and [r["accepted"] for r in tg] == [r["accepted"] for r in to]


def _coop_walk_main():
    """body of test_cooperative_walk_lists_bit_exact (own process: ROLO_KNN_BUDGET is read once per process)"""
    out = []
    cases = [("os1-128", 1, None), ("os1-128", 1, synth.pool_origin(4)), ("os1-64", 1, None), ("os1-128", 3, None), ("vlp16", 1, None)]
    for sensor, stride, origin in cases:   # 16, 16, 8 and 4 wavefronts per workgroup; pool pair 4 holds the heaviest packets of the bench's pool
        src, tgt, _ = synth.dense_pair(sensor, col_stride=stride, origin=origin)
        g = RotVGICP(); g.setResolution(0.5)
        g.setInputTarget(tgt); g.setInputSource(src)
        idx_g, d2_g = g.knn(0)
        idx_o, d2_o = pyorc.knn(src, 20)
        out.append((sensor, stride, src.shape[0], bool(np.array_equal(idx_g, idx_o)), bool(np.array_equal(d2_g, d2_o))))
        g.computeCovariances()   # the pair launch (both clouds in one grid)
        o = pyorc.Reg(pyorc.default_params(voxel_type=1, voxel_resolution=0.5)); o.set_target(tgt); o.set_source(src); o.compute_covariances()
        out.append((sensor, stride, "covs", bool(np.abs(g.getSourceCovariances() - o.source_covs()).max() < 1e-9), bool(np.abs(g.getTargetCovariances() - o.target_covs()).max() < 1e-9)))
        g.close()
    print("COOP", out)
    assert all(a and b for *_, a, b in out), out


@pytest.mark.parametrize("budget", [6, 24])
def test_cooperative_walk_lists_bit_exact(budget):
    """knn_walk_coop_kernel (ROLO_KNN_BUDGET > 0, an opt-in: measured no faster than the plain walk) — a heavy packet's sub-trees stolen by the idle
    wavefronts of its workgroup, walked into fresh lists under the donor's published bound and merged after a barrier: neighbour lists and float
    distances bit-identical to the oracle's at every workgroup shape (4 / 8 / 16 packets), with an early budget (most packets donate) and a late one."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ROLO_KNN_BUDGET=str(budget))
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from tests.test_gpu_registration import _coop_walk_main; _coop_walk_main()" % root],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "COOP" in r.stdout


@pytest.mark.parametrize("sub", [0, 1, 2])
def test_both_walk_kernels_lists_bit_exact(sub):
    """launch_knn_walk picks the k = 20 walk by size: 16 queries x 4 lanes per wavefront for small clouds (knn_walk_sub_kernel: per-sub-lane lists under a shared
    bound, two tree levels per step, merged at the end), 64-query packets for large ones. ROLO_KNN_SUB forces one or the other: both must give the oracle's
    neighbour lists and float distances bit for bit at every size (the cases of the cooperative walk's test: 131 072, a heavy pool pair, 65 536, 43 776, 28 800)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ROLO_KNN_SUB=str(sub)); env.pop("ROLO_KNN_BUDGET", None)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from tests.test_gpu_registration import _coop_walk_main; _coop_walk_main()" % root],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "COOP" in r.stdout


@pytest.mark.parametrize("sub", [0, 2])
def test_forced_walk_kernels_on_degenerate_and_odd_clouds(sub):
    """the default policy sends every small cloud through the four-lanes-per-query walk; the 64-query packets (ROLO_KNN_SUB=0) and the two-lane form (=2) must
    pass the same tie / duplicate / rank-deficient / odd-size cases (own process: the switch is read once)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_registration.py"), "-m", "gpu", "-x", "-q", "-k",
                        "degenerate_geometry or odd_sizes or knn_lists_bit_exact"], env=dict(os.environ, ROLO_KNN_SUB=str(sub)), capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2500:] + r.stderr[-1500:]


def _edge_keys_main():
    """body of the fast-path leg of test_polar_keys_at_planted_bin_edges (own process: ROLO_POLAR_EXACT is read once per process)"""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "polar_edge_points.npz"))
    g = RotVGICP(); g.setPolarResolution(0.175, 0.175, 2.0)
    g.setInputTarget(d["points"]); g.setInputSource(d["points"])
    wrong = int((g.targetVoxelKeys() != d["keys"]).any(axis=1).sum())
    g.so3_linearize(np.eye(4))
    found, keys = g.correspondences()
    print("EDGEDIFF", wrong, "LOOKUPMISS", int((~found[:, 0].astype(bool)).sum()))


def test_polar_keys_at_planted_bin_edges():
    """VERDICT r03 item 6 / SURVEY section 7: 1166 float32 points planted within 5e-13 bins of a theta / phi / r bin edge of the production POLAR grid
    (tests/golden/make_golden_edges.py) — where a libm a few ulp off files a point under the neighbouring key. The map build re-keys such points with the
    correctly rounded atan2 / acos (polar_exact.hpp): keys bit-identical to the oracle's (glibc), through rolo_get_target_voxel_keys AND through the map
    itself (same voxels, same counts). The leg with ROLO_POLAR_EXACT=0 reports how many keys the fast device functions alone get wrong."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = np.load(os.path.join(root, "tests", "golden", "polar_edge_points.npz"))
    pts = d["points"]
    assert pts.shape[0] >= 1000
    src, tgt, cfg = make_pair("vlp16_polar")
    cloud = np.concatenate([pts, tgt[:6000]]).astype(np.float32)   # the planted points inside an ordinary scan
    g = RotVGICP(); g.setPolarResolution(*cfg["polar"])
    g.setInputTarget(cloud); g.setInputSource(src)
    k_o = pyorc.voxel_keys(cloud, 0, polar_res=cfg["polar"])
    assert np.array_equal(k_o[:pts.shape[0]], d["keys"])            # the fixture's keys (authoring container) = this box's oracle
    assert np.array_equal(g.targetVoxelKeys(), k_o)                  # every key, planted or not, bit for bit
    g.buildVoxelMap()
    assert g.numEdgePoints() >= pts.shape[0]                         # all of them were seen as edge points (the statistic stays)
    kg, cg, _, _ = g.voxels()
    o = pyorc.Reg(pyorc.default_params(polar_resolution=cfg["polar"])); o.set_target(cloud); o.set_source(src)
    assert o.compute_covariances() == 0 and o.build_voxelmap() == 0
    ko, co, _, _ = o.voxels()
    og, oo = np.lexsort(kg.T[::-1]), np.lexsort(ko.T[::-1])
    assert np.array_equal(kg[og], ko[oo]) and np.array_equal(cg[og], co[oo])   # the same voxels with the same point counts
    # the pass kernels' LOOKUP of a (transformed) source point goes through the same re-keying: with the planted points as the source at the identity pose
    # every one of them must find the voxel it was filed under as a target point, and the correspondence list equals the oracle's
    g.setInputSource(pts); o.set_source(pts)
    assert o.compute_covariances() == 0
    g.so3_linearize(np.eye(4)); o.so3_linearize(np.eye(4))
    found, keys = g.correspondences()
    s_o, v_o = o.correspondences()
    assert found[:, 0].all() and np.array_equal(keys[:, 0], d["keys"])
    order = np.argsort(s_o, kind="stable")
    assert np.array_equal(np.nonzero(found[:, 0])[0], np.sort(s_o)) and np.array_equal(keys[s_o[order], 0], ko[v_o[order]])
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from tests.test_gpu_registration import _edge_keys_main; _edge_keys_main()" % root],
                       env=dict(os.environ, ROLO_POLAR_EXACT="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    print("fast device atan2 / acos alone among the planted points (wrong keys, then lookups that miss their own voxel):", r.stdout.strip().split("EDGEDIFF")[-1].strip())


@pytest.mark.parametrize("kind", ["sparse", "clumped"])
def test_voxel_map_through_the_workgroup_table(kind):
    """K6's accumulation lets the sums of a workgroup's 256 curve-consecutive points meet in a 128-entry LDS table before they go to the records (voxel_dev.hpp
    accumulate_point_wg). sparse: 8 192 points, (almost) every one a voxel of its own — more distinct voxels per workgroup than the table holds, so runs overflow straight to the
    records; clumped: 16 384 points in 24 voxels — every table entry collects hundreds of points. Both the staged build (buildVoxelMap) and the map a frame builds inside the
    search's launches (register_async: VoxelFuse) against the oracle's map: keys and counts equal, means to 1e-12, covariances to 1e-9."""
    rng = np.random.default_rng(20261001)
    if kind == "sparse":
        n = 8192
        p = rng.uniform(-90.0, 90.0, size=(n, 3))
    else:
        n = 16384
        centres = rng.integers(-20, 20, size=(24, 3)).astype(np.float64) * 0.5 + 0.5   # voxel k covers [k + 0.5, k + 1.5) * leaf: centres sit mid-voxel
        p = centres[rng.integers(0, 24, size=n)] + rng.uniform(-0.2, 0.2, size=(n, 3))
    tgt = np.c_[p, np.zeros(n)].astype(np.float32)
    src = (tgt + np.array([0.01, -0.02, 0.015, 0.0], np.float32)).astype(np.float32)
    cfg = dict(voxel_type=1, leaf=0.5, polar=(0.175, 0.175, 2.0))
    o, g = make_both(src, tgt, cfg)
    assert o.compute_covariances() == 0 and o.build_voxelmap() == 0
    ko, co, mo, vo = o.voxels()
    so = np.lexsort(ko.T[::-1])
    if kind == "sparse":
        assert len(ko) > 0.98 * n
    else:
        assert len(ko) <= 24 * 8
    def check():
        kg, cg, mg, vg = g.voxels()
        sg = np.lexsort(kg.T[::-1])
        assert kg.shape == ko.shape and np.array_equal(kg[sg], ko[so]) and np.array_equal(cg[sg], co[so])
        assert np.abs(mg[sg] - mo[so]).max() < 1e-12 and np.abs(vg[sg] - vo[so]).max() < 1e-9
    g.buildVoxelMap(); check()                       # the staged build: voxel_accum_sorted_kernel
    g2 = RotVGICP(); g2.setResolution(0.5); g2.setInputTarget(tgt); g2.setInputSource(src)
    G = np.zeros(3)
    g2.register_async(None, np.zeros(3), G, G); g2.register_wait()
    g_keep, g = g, g2
    check()                                          # the frame's own build: the tail's accumulation
    g_keep.close(); g2.close()
