"""GPU parity of the front end (K1-K4) against the oracle: integer / index outputs bit-exact, float outputs
bit-exact as well (same fp32 expressions, FMA contraction off on both sides)."""
import numpy as np
import pytest

from oracle import pyorc
from rolo_amd import synth
from rolo_amd.frontend import FrontEnd, front_params, deskew_params
from rolo_amd.rotvgicp import RotVGICP

pytestmark = pytest.mark.gpu

SENSORS = {"vlp16": dict(n_scan=16, horizon_scan=1800), "os1-64": dict(n_scan=64, horizon_scan=1024),
           "os1-128": dict(n_scan=128, horizon_scan=1024), "os1-128x2048": dict(n_scan=128, horizon_scan=2048),
           "wide-32x4096": dict(n_scan=32, horizon_scan=4096)}


def both(sensor, frame, **kw):
    cfg = dict(SENSORS[sensor]); cfg.update(kw)
    fo = pyorc.front_params(**cfg)
    fg = front_params(**cfg)
    ctx = RotVGICP()
    fe = FrontEnd(ctx, fg)
    po = pyorc.project(fo, frame.xyz, frame.ring)
    pg = fe.project(frame.xyz, frame.ring, want_range_mat=True)
    return fo, po, fe, pg


def check_projection(po, pg):
    assert pg["n"] == po["n"]
    assert np.array_equal(pg["start_ring"], po["start_ring"]) and np.array_equal(pg["end_ring"], po["end_ring"])
    assert np.array_equal(pg["point_col_ind"], po["point_col_ind"])       # bit-exact indices
    assert np.array_equal(pg["point_range"], po["point_range"])           # fp32, same expression
    assert np.array_equal(pg["extracted"], po["extracted"])
    assert np.array_equal(pg["range_mat"], po["range_mat"])


def check_features(fo, po, fe, pg):
    eo = pyorc.extract_features(fo, po)
    eg = fe.extract(pg["n"], debug=True)
    assert np.array_equal(eg["curvature"], eo["curvature"])
    assert np.array_equal(eg["picked"], eo["picked"])
    assert np.array_equal(eg["label"], eo["label"])
    assert eg["corner"].shape == eo["corner"].shape and np.array_equal(eg["corner"], eo["corner"])
    assert eg["surface"].shape == eo["surface"].shape and np.array_equal(eg["surface"], eo["surface"])
    return eo, eg


@pytest.mark.parametrize("sensor", ["vlp16", "os1-64", "os1-128", "os1-128x2048", "wide-32x4096"])
def test_projection_and_features_match_oracle(sensor):
    f0, f1, _ = synth.make_pair(sensor)
    for fr in (f0, f1):
        fo, po, fe, pg = both(sensor, fr)
        check_projection(po, pg)
        eo, eg = check_features(fo, po, fe, pg)
        assert eo["corner"].shape[0] > 0 and eo["surface"].shape[0] > 0


def test_duplicate_pixels_first_point_wins_and_filters():
    f0, _, _ = synth.make_pair("vlp16")
    rng = np.random.default_rng(3)
    # shuffle, duplicate a slice (same pixels twice), add out-of-range / bad-ring points
    idx = rng.permutation(f0.xyz.shape[0])
    xyz = np.concatenate([f0.xyz[idx], f0.xyz[idx[:5000]] * 1.01, np.array([[0.5, 0.2, 0.1], [1500.0, 0, 0]], np.float32)])
    ring = np.concatenate([f0.ring[idx], f0.ring[idx[:5000]], np.array([3, 4], np.uint16)])
    ring[100:110] = 99  # invalid ring
    fr = synth.Frame(xyz.astype(np.float32), np.zeros(len(xyz), np.float32), ring.astype(np.uint16), np.zeros(len(xyz), np.float32), 16, 1800)
    fo, po, fe, pg = both("vlp16", fr)
    check_projection(po, pg)
    check_features(fo, po, fe, pg)


def test_downsample_rate_and_empty_rings():
    f0, _, _ = synth.make_pair("os1-64")
    fo, po, fe, pg = both("os1-64", f0, downsample_rate=2)
    check_projection(po, pg)
    assert np.all((po["end_ring"] - po["start_ring"])[1::2] == -10)  # odd rings are empty
    check_features(fo, po, fe, pg)


def test_sparse_and_tiny_inputs():
    f0, _, _ = synth.make_pair("vlp16", col_stride=16)  # 112 points per ring
    fo, po, fe, pg = both("vlp16", f0)
    check_projection(po, pg)
    check_features(fo, po, fe, pg)
    fr = synth.Frame(f0.xyz[:7], f0.intensity[:7], f0.ring[:7], f0.time[:7], 16, 1800)
    fo, po, fe, pg = both("vlp16", fr)
    check_projection(po, pg)
    check_features(fo, po, fe, pg)


@pytest.mark.parametrize("edge,surf", [(0.1, 0.02), (0.02, 0.02), (0.003, 0.002), (0.05, 0.1)])
def test_thresholds_change_the_selection(edge, surf):
    """other thresholds, chosen to leave the fast path of the pick stages (one fixed point over the whole ring): many corner candidates — sectors
    with more than the 20 corner picks the reference's walk stops at are redone stage by stage — and an edge threshold BELOW the surface one
    (a cell can be both kinds of candidate: staged from the start)"""
    f0, _, _ = synth.make_pair("os1-64")
    fo, po, fe, pg = both("os1-64", f0, edge_threshold=edge, surf_threshold=surf, odometry_surf_leaf_size=0.2)
    check_projection(po, pg)
    eo, eg = check_features(fo, po, fe, pg)
    if edge <= 0.02:
        assert eo["corner"].shape[0] >= 64 * 6 * 10   # the cap is what bounds the corner count here


def test_deskew_matches_oracle():
    """rolo_front_set_deskew = ImageProjection::deskewPoint: every index output is untouched (range, pixel and curvature come
    from the raw point), the stored coordinates are the oracle's up to the last bit of device sinf / cosf."""
    from rolo_amd.frontend import deskew_params, odom_increment
    from rolo_amd.rotvgicp import RotVGICP
    cfg = dict(n_scan=16, horizon_scan=1800)
    fr = synth.make_frame("vlp16", np.eye(3), np.zeros(3), synth.SEED)
    n = fr.xyz.shape[0]
    rel_time = (np.arange(n, dtype=np.float64) / n * 0.1).astype(np.float32)     # firing order over a 0.1 s scan
    inc_o = pyorc.odom_increment([0, 0, 0, 0, 0, 0], [0.31, 0.04, 0.01, 0.004, -0.006, 0.035])
    inc_g = odom_increment([0, 0, 0, 0, 0, 0], [0.31, 0.04, 0.01, 0.004, -0.006, 0.035])
    assert np.array_equal(inc_o, inc_g)
    fo = pyorc.front_params(**cfg); fg = front_params(**cfg)
    po = pyorc.project(fo, fr.xyz, fr.ring, rel_time, pyorc.deskew(inc_o[3:], 0.1, 0.0987))
    eo = pyorc.extract_features(fo, po)
    g = RotVGICP(); fe = FrontEnd(g, fg)
    fe.setDeskew(deskew_params(inc_g[3:], 0.1, 0.0987), rel_time)
    pg = fe.project(fr.xyz, fr.ring); eg = fe.extract(pg["n"], debug=True)
    assert pg["n"] == po["n"] and np.array_equal(pg["point_col_ind"], po["point_col_ind"]) and np.array_equal(pg["point_range"], po["point_range"])
    assert np.array_equal(eg["label"], eo["label"]) and np.array_equal(eg["picked"], eo["picked"])
    raw = pyorc.project(fo, fr.xyz, fr.ring)
    assert np.abs(po["extracted"][:, :3] - raw["extracted"][:, :3]).max() > 0.05          # the de-skew really moved the points
    assert np.abs(pg["extracted"][:, :3] - po["extracted"][:, :3]).max() <= 4e-6          # float ulps at <= 100 m
    assert np.array_equal(pg["extracted"][:, 3], po["extracted"][:, 3])                   # intensity = ring * raw z
    assert eg["corner"].shape == eo["corner"].shape and np.abs(eg["corner"] - eo["corner"]).max() <= 4e-6
    assert eg["surface"].shape == eo["surface"].shape and np.abs(eg["surface"] - eo["surface"]).max() <= 1e-5
    # armed for one projection only; a disabled block is the plain path
    p2 = fe.project(fr.xyz, fr.ring)
    assert np.array_equal(p2["extracted"], raw["extracted"])
    fe.setDeskew(deskew_params(inc_g[3:], 0.1, 0.0987, enabled=False), rel_time)
    assert np.array_equal(fe.project(fr.xyz, fr.ring)["extracted"], raw["extracted"])


def test_nan_points_are_dropped_like_on_the_reference_cpu():
    """Non-dense clouds are refused by the node (imageProjection.cpp:226-231); if NaNs arrive anyway the reference's column
    index is int(NaN) = INT_MIN on x86 and the point is dropped — same here (the device's int(NaN) would be 0)."""
    from rolo_amd.rotvgicp import RotVGICP
    cfg = dict(n_scan=16, horizon_scan=1800)
    fr = synth.make_frame("vlp16", np.eye(3), np.zeros(3), synth.SEED)
    xyz = np.array(fr.xyz, np.float32); ring = np.array(fr.ring, np.uint16)
    bad = np.arange(0, xyz.shape[0], 97)
    xyz[bad[0::3], 0] = np.nan; xyz[bad[1::3], 1] = np.nan; xyz[bad[2::3]] = np.nan
    fo = pyorc.front_params(**cfg); fg = front_params(**cfg)
    po = pyorc.project(fo, xyz, ring); eo = pyorc.extract_features(fo, po)
    g = RotVGICP(); fe = FrontEnd(g, fg)
    pg = fe.project(xyz, ring); eg = fe.extract(pg["n"])
    assert pg["n"] == po["n"] and np.array_equal(pg["extracted"], po["extracted"]) and np.array_equal(pg["point_col_ind"], po["point_col_ind"])
    assert not np.isnan(pg["extracted"]).any()
    assert np.array_equal(eg["corner"], eo["corner"]) and np.array_equal(eg["surface"], eo["surface"])


@pytest.mark.parametrize("name", ["front_vlp16", "front_os64"])
def test_front_end_against_committed_golden(name):
    """K1-K4 through the C ABI against tests/golden/front_*.npz (outputs of the independent numpy twin, oracle/twin_front.py):
    thinned frames with duplicates, out-of-range points, an unknown ring and NaNs. Bit-exact."""
    import os
    from rolo_amd.rotvgicp import RotVGICP
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    fg = front_params(n_scan=int(z["n_scan"]), horizon_scan=int(z["horizon_scan"]))
    g = RotVGICP(); fe = FrontEnd(g, fg)
    pg = fe.project(z["xyz"], z["ring"])
    owner = z["owner"]
    ext = np.zeros((owner.size, 4), np.float32)
    ext[:, :3] = z["xyz"][owner]; ext[:, 3] = z["ring"][owner].astype(np.float32) * z["xyz"][owner, 2]
    assert pg["n"] == owner.size and np.array_equal(pg["extracted"], ext)
    for k in ("point_col_ind", "point_range", "start_ring", "end_ring"):
        assert np.array_equal(pg[k], z[k]), k
    eg = fe.extract(pg["n"], debug=True)
    assert np.array_equal(eg["curvature"], z["curvature"])
    assert np.array_equal(eg["picked"], z["picked"].astype(np.int32)) and np.array_equal(eg["label"], z["label"].astype(np.int32))
    assert np.array_equal(eg["corner"], z["corner"]) and np.array_equal(eg["surface"], z["surface"])


def test_deskew_with_azimuth_times():
    """rel_time == NULL: the per-point times are interpolated from the azimuth on the device as deskewCloudInfo does for clouds
    without a time field (timeFlag == -1) — the oracle's serial loop gives the same times and the same de-skewed cloud."""
    from rolo_amd.frontend import deskew_params
    from rolo_amd.rotvgicp import RotVGICP
    cfg = dict(n_scan=16, horizon_scan=1800)
    fr = synth.make_frame("vlp16", synth.rpy_to_R(0, 0, 0.4), np.zeros(3), synth.SEED)
    order = np.argsort(-np.arctan2(fr.xyz[:, 1], fr.xyz[:, 0]), kind="stable")      # one sweep, like a spinning sensor
    xyz = np.asarray(fr.xyz, np.float32)[order]; ring = np.asarray(fr.ring, np.uint16)[order]
    fo = pyorc.front_params(**cfg); fg = front_params(**cfg)
    inc = [0.004, -0.006, 0.035]
    times = pyorc.azimuth_times(xyz, 0.1)
    po = pyorc.project(fo, xyz, ring, times, pyorc.deskew(inc, 0.1, 0.0987))
    g = RotVGICP(); fe = FrontEnd(g, fg)
    fe.setDeskewFromCloud(deskew_params(inc, 0.1, 0.0987))
    pg = fe.project(xyz, ring)
    raw = pyorc.project(fo, xyz, ring)
    assert pg["n"] == po["n"] and np.array_equal(pg["point_col_ind"], po["point_col_ind"]) and np.array_equal(pg["point_range"], po["point_range"])
    assert np.abs(po["extracted"][:, :3] - raw["extracted"][:, :3]).max() > 0.05
    assert np.abs(pg["extracted"][:, :3] - po["extracted"][:, :3]).max() <= 6e-6


def test_load_projection_refuses_indices_that_would_leave_the_arrays():
    """rolo_front_load_projection (the feature node's input when it runs as its own process): startRingIndex / endRingIndex / pointColInd are
    used as array indices on the device — a malformed rolo/cloud_info is ROLO_EINVAL, not an out-of-bounds access."""
    import ctypes as C
    from rolo_amd._lib import lib, FrontParams
    from rolo_amd.rotvgicp import RotVGICP
    g = RotVGICP()
    P = FrontParams(); lib().rolo_front_default_params(C.byref(P)); P.n_scan = 16; P.horizon_scan = 1800
    n = 1000
    ext = np.zeros((n, 4), np.float32); col = (np.arange(n) % 1800).astype(np.int32); rng = np.ones(n, np.float32)
    cnt = np.linspace(0, n, 17).astype(np.int32)
    start = (cnt[:-1] - 1 + 5).astype(np.int32); end = (cnt[1:] - 1 - 5).astype(np.int32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)

    def call(s_, e_, c_):
        return lib().rolo_front_load_projection(g._h, C.byref(P), ext.ctypes.data_as(fp), c_.ctypes.data_as(ip), rng.ctypes.data_as(fp), s_.ctypes.data_as(ip), e_.ctypes.data_as(ip), n)
    assert call(start, end, col) == 0
    bad = start.copy(); bad[3] = 10 ** 6
    assert call(bad, end, col) == -1
    bad = end.copy(); bad[15] = n + 50
    assert call(start, bad, col) == -1
    bad = end.copy(); bad[2] = -100
    assert call(start, bad, col) == -1
    bad = col.copy(); bad[17] = 1800
    assert call(start, end, bad) == -1
    bad = col.copy(); bad[0] = -1
    assert call(start, end, bad) == -1
    g.close()


def test_recycled_context_forgets_its_previous_owners_projection():
    """rolo_ctx_release -> rolo_ctx_acquire hands the SAME context back as a fresh object: the previous owner's projection must not be there for a staged
    rolo_extract_features to run on, its armed de-skew must not fire on the new owner's first frame, and the per-object counters start at zero"""
    import ctypes as C
    from rolo_amd._lib import lib, RoloError
    from rolo_amd.rotvgicp import RotVGICP
    L = lib()
    L.rolo_ctx_pool_clear()
    h = C.c_void_p(); assert L.rolo_ctx_acquire(0, C.byref(h)) == 0
    first = h.value
    fr = synth.make_frame("vlp16", np.eye(3), np.zeros(3), synth.SEED)
    cfg = dict(n_scan=16, horizon_scan=1800)
    a = RotVGICP(0, _borrowed=h.value)
    fe = FrontEnd(a, front_params(**cfg))
    pg = fe.project(fr.xyz, fr.ring)                     # projected, never extracted
    fe.setDeskewFromCloud(deskew_params((0.01, 0.02, 0.03)))   # ... and a de-skew armed for a frame that never comes
    L.rolo_ctx_release(h)
    h2 = C.c_void_p(); assert L.rolo_ctx_acquire(0, C.byref(h2)) == 0
    assert h2.value == first
    b = RotVGICP(0, _borrowed=h2.value)
    fe2 = FrontEnd(b, front_params(**cfg))
    with pytest.raises(RoloError) as ei:
        fe2.extract(pg["n"])
    assert ei.value.code == -5                           # ROLO_ESTATE: this object has not projected anything
    assert b.counters()["frames"] == 0
    p2 = fe2.project(fr.xyz, fr.ring); e2 = fe2.extract(p2["n"])
    fo = pyorc.front_params(**cfg)
    eo = pyorc.extract_features(fo, pyorc.project(fo, fr.xyz, fr.ring))   # NO de-skew: the previous owner's armed one is gone
    assert np.array_equal(p2["extracted"], pg["extracted"]) and np.array_equal(e2["corner"], eo["corner"]) and np.array_equal(e2["surface"], eo["surface"])
    L.rolo_ctx_release(h2); L.rolo_ctx_pool_clear()
