"""GPU: SURVEY 8f.4 — the back end's scan-to-submap optimisation (rolo_scan2map_optimize) against the C++ oracle (oracle/rolo_oracle_backend.cpp: the
reference's loops with cv::eigen / colPivHouseholderQr restated in float — the same algorithms the kernel restates, written independently) on feature clouds
the oracle's front end extracts from synthetic frames: the selection flags of the last iteration BIT-IDENTICAL, their point-to-line / point-to-plane
coefficients to float rounding, the same iterations and the optimised pose; plus the independent numpy twin (oracle/twin_backend.py, library eigh / lstsq:
pose <= 1e-4 m, <= 1e-5 rad) and the ground truth of the synthetic trajectory."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import pyorc, twin_backend
from rolo_amd import synth
from rolo_amd.backend import Scan2Map

pytestmark = pytest.mark.gpu


def features(sensor, cfg, R, t, seed):
    fo = pyorc.front_params(**cfg)
    fr = synth.make_frame(sensor, R, t, seed)
    e = pyorc.extract_features(fo, pyorc.project(fo, fr.xyz, fr.ring))
    return e["corner"], e["surface"]


def to_world(pts, R, t):
    o = pts.copy(); o[:, :3] = (pts[:, :3].astype(np.float64) @ R.T + t).astype(np.float32)
    return o


@pytest.mark.parametrize("sensor,cfg", [("vlp16", dict(n_scan=16, horizon_scan=1800)), ("os1-64", dict(n_scan=64, horizon_scan=1024))])
def test_scan2map_matches_twin_and_recovers_the_pose(sensor, cfg):
    # sub-map: the features of two key frames in the map frame; scan: a third frame at a known pose
    poses = [(np.eye(3), np.zeros(3)), (synth.rpy_to_R(0.002, -0.003, 0.03), np.array([0.4, 0.03, 0.0])),
             (synth.rpy_to_R(0.004, -0.002, 0.06), np.array([0.8, 0.08, 0.01]))]
    mc, ms = [], []
    for k in range(2):
        c, s = features(sensor, cfg, *poses[k], synth.SEED + k)
        mc.append(to_world(c, *poses[k])); ms.append(to_world(s, *poses[k]))
    mc = np.concatenate(mc); ms = np.concatenate(ms)
    corner, surf = features(sensor, cfg, *poses[2], synth.SEED + 2)
    R2, t2 = poses[2]
    rpy = Rotation.from_matrix(R2).as_euler("xyz")
    truth = np.concatenate([rpy, t2]).astype(np.float32)
    guess = (truth + np.array([0.004, -0.003, 0.01, 0.06, -0.04, 0.02], np.float32)).astype(np.float32)

    g = Scan2Map()
    tf_g, sel_g, co_g = g.scan2MapOptimization(corner, surf, mc, ms, guess, want_debug=True)
    tf_t, st_t, sel_t, co_t = twin_backend.scan2map(corner, surf, mc, ms, guess)
    st = g.last_stats
    # ---- the C++ oracle: the standard of the other rows ----
    tf_o, st_o, sel_o, co_o = pyorc.scan2map(corner, surf, mc, ms, guess)
    assert (st.skipped, st.iterations, st.converged, st.degenerate, st.n_selected) == tuple(st_o[k] for k in ("skipped", "iterations", "converged", "degenerate", "n_selected"))
    assert np.array_equal(sel_g, sel_o)                               # laserCloudOri*Flag of the last iteration: bit-identical
    assert np.abs(co_g - co_o).max() <= 1e-6                          # coeffSel: float rounding (the J^T J rows are summed in another order: the poses differ in the last float bits)
    assert np.abs(tf_g - tf_o).max() <= 2e-6
    # ---- the independent twin ----
    assert st.skipped == 0 and st_t["skipped"] == 0
    assert st.converged == st_t["converged"] == 1 and st.degenerate == st_t["degenerate"]
    assert abs(st.iterations - st_t["iterations"]) <= 1
    # the last iteration's association: the same points selected but for threshold-borderline cases (float fits on two different
    # eigen / least-squares routines), the same coefficients on the common ones
    both = sel_g & sel_t
    assert (sel_g != sel_t).mean() < 5e-3 and both.sum() > 0.5 * len(sel_t)
    # float covariances of five nearly collinear / coplanar points are ill-conditioned: the two fits agree to ~1e-4 typically, a few
    # near-degenerate neighbourhoods differ more; what is held tight is the pose below
    dco = np.abs(co_g[both] - co_t[both]).max(axis=1)
    assert np.median(dco) < 1e-3 and np.percentile(dco, 99) < 5e-2
    assert np.abs(tf_g[3:] - tf_t[3:]).max() <= 1e-4 and np.abs(tf_g[:3] - tf_t[:3]).max() <= 1e-5
    # and it is the right pose: the perturbation is removed to the level the synthetic noise allows
    assert np.abs(tf_g[3:] - truth[3:]).max() < 0.03 and np.abs(tf_g[:3] - truth[:3]).max() < 3e-3
    assert np.abs(guess[3:] - truth[3:]).max() > 0.05

    # the sub-map resident in the context (rolo_scan2map_set_submap): the same bits as the one-call form
    g.setSubmap(mc, ms)
    for _ in range(2):
        tf_r, sel_r, co_r = g.scan2MapOptimization(corner, surf, None, None, guess, want_debug=True)
        assert np.array_equal(tf_r, tf_g) and np.array_equal(sel_r, sel_g) and np.array_equal(co_r, co_g)
    # too few features: nothing happens (backMapping.cpp:689, 708)
    tf_s = g.scan2MapOptimization(corner[:5], surf, mc, ms, guess)
    assert g.last_stats.skipped == 1 and np.array_equal(tf_s, guess)
    g.close()


def test_resident_submap_call_order():
    """rolo_scan2map_optimize(..., NULL, 0, NULL, 0, ...) needs a preceding rolo_scan2map_set_submap (ROLO_ESTATE otherwise); a sub-map without five points in one
    of its clouds cannot answer a 5-NN query: skipped = 2, the pose untouched"""
    from rolo_amd._lib import RoloError
    corner, surf = features("vlp16", dict(n_scan=16, horizon_scan=1800), np.eye(3), np.zeros(3), synth.SEED)
    guess = np.zeros(6, np.float32)
    g = Scan2Map()
    with pytest.raises(RoloError) as ei:
        g.scan2MapOptimization(corner, surf, None, None, guess)
    assert ei.value.code == -5
    g.setSubmap(corner[:3], surf)
    tf = g.scan2MapOptimization(corner, surf, None, None, guess)
    assert g.last_stats.skipped == 2 and np.array_equal(tf, guess)
    g.setSubmap(corner, surf)
    tf = g.scan2MapOptimization(corner, surf, None, None, guess)    # the scan against itself: aligned up to the fits through five neighbours
    assert g.last_stats.skipped == 0 and g.last_stats.iterations >= 1 and np.abs(tf).max() < 1e-2, (g.last_stats.n_selected, g.last_stats.converged)
    g.close()


def test_resident_submap_goes_stale_with_the_contexts_clouds():
    """the resident sub-map's trees are raw pointers into the context's own source / target clouds (advisor, round 4): a later upload into the same context, or a
    trip through the context pool, must turn rolo_scan2map_optimize(NULL, NULL) into ROLO_ESTATE instead of a walk over overwritten / freed memory; a failed or
    too-small rolo_scan2map_set_submap leaves nothing resident either"""
    from rolo_amd._lib import RoloError, lib
    import ctypes as C
    corner, surf = features("vlp16", dict(n_scan=16, horizon_scan=1800), np.eye(3), np.zeros(3), synth.SEED)
    guess = np.zeros(6, np.float32)
    g = Scan2Map()
    g.setSubmap(corner, surf)
    tf0 = g.scan2MapOptimization(corner, surf, None, None, guess)
    assert g.last_stats.skipped == 0
    # (1) another cloud lands in the context
    g.reg.setInputSource(np.ascontiguousarray(surf[:4096]))
    with pytest.raises(RoloError) as ei:
        g.scan2MapOptimization(corner, surf, None, None, guess)
    assert ei.value.code == -5 and "overwritten" in str(ei.value)
    # ... and the next call without a new sub-map is refused as well (nothing is resident any more), a new one works and gives the same bits
    with pytest.raises(RoloError):
        g.scan2MapOptimization(corner, surf, None, None, guess)
    g.setSubmap(corner, surf)
    assert np.array_equal(g.scan2MapOptimization(corner, surf, None, None, guess), tf0)
    # (2) through the pool: the next owner of the context does not inherit the sub-map
    fp = C.POINTER(C.c_float)
    lib().rolo_ctx_pool_clear()   # so that the context released below is the one the next acquire hands out
    h = C.c_void_p()
    assert lib().rolo_ctx_acquire(0, C.byref(h)) == 0
    a = [np.ascontiguousarray(x, np.float32).reshape(-1, 4) for x in (corner, surf)]
    assert lib().rolo_scan2map_set_submap(h, a[0].ctypes.data_as(fp), a[0].shape[0], a[1].ctypes.data_as(fp), a[1].shape[0]) == 0
    lib().rolo_ctx_release(h)
    h2 = C.c_void_p()
    assert lib().rolo_ctx_acquire(0, C.byref(h2)) == 0 and h2.value == h.value   # the pool hands the same object out again
    tf = guess.copy()
    rc = lib().rolo_scan2map_optimize(h2, a[0].ctypes.data_as(fp), a[0].shape[0], a[1].ctypes.data_as(fp), a[1].shape[0], None, 0, None, 0, tf.ctypes.data_as(fp), 10, 100, None, None, None)
    assert rc == -5
    lib().rolo_ctx_release(h2)
    g.close()


@pytest.mark.parametrize("switch", ["ROLO_S2M_SUB=1", "ROLO_S2M_SUB=8", "ROLO_S2M_CAP=0", "ROLO_S2M_WIDE=0", "ROLO_S2M_WIDE=6", "ROLO_S2M_PACKETS=0"])
def test_association_kernel_variants_keep_the_oracles_flags(switch):
    """the association kernel's earlier forms and A/B switches (64-feature packets, 8 lanes per feature, no radius cap, binary steps, paired leaf fetches, one walk
    per lane) are all exact searches: the parity test above — selection flags bit-identical to the C++ oracle's, same iterations, same pose — must pass with
    each of them (own process: the switches are read once)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    k, v = switch.split("=")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_backend.py"), "-m", "gpu", "-x", "-q", "-k", "matches_twin and vlp16"],
                       env=dict(os.environ, **{k: v}), capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2500:] + r.stderr[-1500:]
