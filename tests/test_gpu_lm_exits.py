"""GPU parity of the LM / GN drivers' EXITS and knobs (K12: lsq_registration_impl.hpp:55-80, 84-148, 152-179, 208-335) against the oracle.

Two kinds of test:
* scripted — the controller kernels of passes.hip and the oracle's drivers are fed the SAME scripted evaluations (the linearisation of outer iteration o,
  the trial cost of (o, trial t)): every branch — rejected and not converged, "lm not converged!!", the iteration cap, rejected-but-converged, NaN and
  infinite gain ratios, a trial accepted on a cost-only pass, zero iteration budgets — is reached with inputs that are the same bits on both sides, so the
  decisions, the iteration counts, the damping and the returned pose can be compared without rounding noise deciding a branch;
* real clouds — the exits that real data reaches robustly (the translation stage rejects significantly on every frame: SURVEY Q2), the iteration caps, the
  knobs away from their defaults (rotation / transformation epsilon, initial lambda factor, lm_max_iterations, max_iterations, a 4 degree guess) and a crafted
  one-point-per-voxel scene whose first gain ratio is 0 / 0.
Every case runs the pass + controller launches and, on real clouds, the one-launch-per-trial form and the one-launch-per-frame form (the resident LM kernel) too
(ROLO_LM_SPEC_LIN at its default)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyorc
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP, LSQ_OPTIMIZER_TYPE

pytestmark = pytest.mark.gpu

G = -np.asarray(synth.PREV_STEP_T)
L0 = G * 0.97
SO3, LM6, GN = LSQ_OPTIMIZER_TYPE.SO3_LevenbergMarquardt, LSQ_OPTIMIZER_TYPE.LevenbergMarquardt, LSQ_OPTIMIZER_TYPE.GaussNewton
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from lm_scenarios import make_script, ROT_SCENARIOS, TRANS_SCENARIOS, EXPECT_ROT, EXPECT_TRANS, BIG, TBIG


def both_scripted(optimizer, params, guess=None):
    kw = dict(optimizer=int(optimizer)); kw.update(params)
    o = pyorc.Reg(pyorc.default_params(**kw))
    g = RotVGICP()
    g.setOptimizerType(int(optimizer))
    for k, v in params.items():
        setattr(g._p, k, v)
    g._push()
    return o, g


def same(a, b, rel):
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    if math.isinf(a) or math.isinf(b):
        return a == b
    return abs(a - b) <= rel * max(abs(a), abs(b), 1e-300)


def compare_traces(tr_o, tr_g, stage):
    a = [r for r in tr_o if r["stage"] == stage]; b = [r for r in tr_g if r["stage"] == stage]
    assert len(a) == len(b), (a, b)
    for ro, rg in zip(a, b):
        assert (ro["outer"], ro["trial"], ro["accepted"]) == (rg["outer"], rg["trial"], rg["accepted"]), (ro, rg)
        assert same(ro["y0"], rg["y0"], 0) and same(ro["yi"], rg["yi"], 0), (ro, rg)            # scripted: the same bits
        assert same(ro["lam"], rg["lam"], 1e-13) and same(ro["rho"], rg["rho"], 1e-11) and same(ro["dnorm"], rg["dnorm"], 1e-11), (ro, rg)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("optimizer", [SO3, LM6])
@pytest.mark.parametrize("name", sorted(ROT_SCENARIOS))
def test_scripted_rotation_stage_exits_match_oracle(name, optimizer, generic):
    params, outers = ROT_SCENARIOS[name]
    dof = 3 if optimizer == SO3 else 6
    script = make_script(dof, outers)
    guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.01, -0.02, 0.07); guess[:3, 3] = (0.3, -0.1, 0.05)
    o, g = both_scripted(optimizer, params)
    o.set_script(*script)
    rc_o, Tf_o, Td_o, it_o, cv_o = o.align(guess)
    rc_g = g.script_align(script, guess, generic_ctrl=generic)
    st = g.last_stats
    assert rc_g == 0
    assert (st.lm_failed, st.n_outer, bool(st.converged)) == (rc_o, it_o, cv_o), (name, (st.lm_failed, st.n_outer, st.converged), (rc_o, it_o, cv_o))
    assert np.abs(g.final_transformation_d - Td_o).max() < 1e-12
    compare_traces(o.trace(), g.trace(), 0)
    # what each scenario is there for really happened (a script that drifts into another branch would still "match"): lm_scenarios.EXPECT_ROT, pinned on the CPU tier too
    acc = [r["accepted"] for r in g.trace()]
    assert (st.lm_failed, st.n_outer, bool(st.converged), acc) == EXPECT_ROT[name][dof]
    if st.lm_failed and st.n_outer == 1 or st.n_outer == 0:
        assert np.array_equal(g.final_transformation_d, guess.astype(np.float64))      # "result still returned": the guess, untouched
    if name == "nan_gain_ratio_is_accepted":
        assert math.isnan(g.trace()[0]["rho"])
    if name in ("minus_inf_gain_ratio", "plus_inf_gain_ratio"):
        assert g.trace()[0]["rho"] == (-math.inf if name[0] == "m" else math.inf)
    # the device's own count of cost-only passes (rolo_stats::n_cost_only): one per trial that followed a rejection
    assert st.n_cost_only == sum(1 for x, y in zip(acc, acc[1:]) if x != 1) if os.environ.get("ROLO_LM_SPEC_LIN", "1") != "0" else st.n_cost_only == 0
    g.close()


@pytest.mark.parametrize("name", ["accept_until_converged", "iteration_cap", "no_iteration_budget", "tight_epsilons"])
def test_scripted_gauss_newton(name):
    params, outers = ROT_SCENARIOS[name]
    script = make_script(6, outers)
    o, g = both_scripted(GN, params)
    o.set_script(*script)
    rc_o, _, Td_o, it_o, cv_o = o.align(None)
    assert g.script_align(script, None) == 0
    st = g.last_stats
    assert (st.lm_failed, st.n_outer, bool(st.converged)) == (rc_o, it_o, cv_o)
    assert np.abs(g.final_transformation_d - Td_o).max() < 1e-12
    a, b = o.trace(), g.trace()
    assert [(r["outer"], r["accepted"]) for r in a] == [(r["outer"], r["accepted"]) for r in b]
    g.close()


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("name", sorted(TRANS_SCENARIOS))
def test_scripted_translation_stage_exits_match_oracle(name, generic):
    params, outers = TRANS_SCENARIOS[name]
    script = make_script(6, outers, seed=11)
    rot = make_script(3, ROT_SCENARIOS["iteration_cap_one"][1])   # one rotation iteration first: it leaves the correspondence count computeTranslation divides by
    o, g = both_scripted(SO3, dict())
    o.set_script(*rot)
    assert o.align(None)[0] == 0 and g.script_align(rot, None) == 0
    o.set_driver_params(**params)
    for k, v in params.items():
        setattr(g._p, k, v)
    g._push()
    start = np.array([0.01, -0.02, 0.005])
    o.set_script(*script); o.clear_trace()
    rc_o, t_o, it_o = o.compute_translation(start, G, L0)
    rc_g, t_g = g.script_translation(script, start, G, L0, generic_ctrl=generic)
    st = g.last_translation_stats
    assert rc_g == 0
    assert (st.lm_failed, st.n_outer) == (rc_o, it_o), (name, (st.lm_failed, st.n_outer), (rc_o, it_o))
    assert np.abs(t_g - t_o).max() < 1e-13
    compare_traces(o.trace(), g.trace(), 1)
    acc = [r["accepted"] for r in g.trace() if r["stage"] == 1]
    assert (st.lm_failed, st.n_outer, acc) == EXPECT_TRANS[name]
    if st.lm_failed and st.n_outer == 1 or st.n_outer == 0:
        assert np.array_equal(t_g, start)
    g.close()


def test_scripted_empty_linearisation_is_an_error_code():
    """a linearisation without correspondences ends the registration with ROLO_ENOCORR (the reference divides by zero and returns NaN)"""
    script = make_script(3, [dict(step=BIG, y0=1e6, trials=[4e5], n=0)])
    g = RotVGICP()
    assert g.script_align(script, None) == -4 and g.last_stats.lm_failed == 1
    # ... and after a budget of zero iterations computeTranslation has nothing to work on
    g.setMaximumIterations(0)
    assert g.script_align(make_script(3, [dict(step=BIG, y0=1e6, trials=[4e5])]), None) == 0 and g.last_stats.n_outer == 0
    rc, _ = g.script_translation(make_script(6, [dict(step=TBIG, y0=1e6)]), np.zeros(3), G, L0)
    assert rc == -4
    g.close()


# ---- real clouds ------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def clouds():
    src, tgt, _ = synth.dense_pair("vlp16", col_stride=2)
    return src, tgt


def make_real(src, tgt, fused, **kw):
    polar = (0.175, 0.175, 2.0)
    o = pyorc.Reg(pyorc.default_params(polar_resolution=polar, voxel_type=0, **kw))
    o.set_target(tgt); o.set_source(src)
    g = RotVGICP(); g.setPolarResolution(*polar)
    for k, v in kw.items():
        setattr(g._p, k, v)
    g._p.fused_lm = int(fused); g._push()
    g.setInputTarget(tgt); g.setInputSource(src)
    return o, g


def significant(r):
    return abs(r["y0"] - r["yi"]) > 1e-7 * abs(r["y0"])


def check_real(o, g, guess=None, trans_start=np.zeros(3), ct_lambda=0.3, expect_rot=None, expect_trans=None):
    rc_o, Tf_o, Td_o, it_o, cv_o = o.align(guess)
    g.align(guess)
    st = g.last_stats
    assert (st.lm_failed, st.n_outer, bool(st.converged)) == (rc_o, it_o, cv_o)
    if expect_rot is not None:
        assert (st.lm_failed, st.n_outer, bool(st.converged)) == expect_rot
    assert np.abs(g.final_transformation_d - Td_o).max() < 1e-8
    rc_t, t_o, tit_o = o.compute_translation(trans_start, G, L0, ct_lambda=ct_lambda)
    t_g = g.computeTranslation(trans_start, G, L0, ct_lambda=ct_lambda)
    ts = g.last_translation_stats
    assert (ts.lm_failed, ts.n_outer) == (rc_t, tit_o)
    if expect_trans is not None:
        assert (ts.lm_failed, ts.n_outer) == expect_trans
    assert np.abs(t_g - t_o).max() < 1e-8
    tr_o, tr_g = o.trace(), g.trace()
    for stage in (0, 1):
        a = [r for r in tr_o if r["stage"] == stage]; b = [r for r in tr_g if r["stage"] == stage]
        # every record whose decision is not rounding noise: same decision, same numbers
        for ro, rg in zip(a, b):
            if not significant(ro):
                break
            assert (ro["outer"], ro["trial"], ro["accepted"]) == (rg["outer"], rg["trial"], rg["accepted"]), (ro, rg)
            assert same(ro["y0"], rg["y0"], 1e-9) and same(ro["yi"], rg["yi"], 1e-9) and same(ro["lam"], rg["lam"], 1e-6), (ro, rg)
    return tr_g


@pytest.mark.parametrize("fused", [0, 1, 2])   # pass + controller launches / one launch per trial / one launch per frame (the resident kernel)
@pytest.mark.parametrize("lm_max", [1, 2, 4, 7, 8])
def test_translation_lm_failure_on_real_clouds(clouds, fused, lm_max):
    """the translation stage's second outer iteration rejects seven trials in a row on this pair (the as-written CT term; every rejection is 0.3 % of the cost,
    not noise): with lm_max_iterations <= 7 both sides give up ("lm not converged!!", :66-69) in the second iteration started and return the first iteration's result"""
    src, tgt = clouds
    o, g = make_real(src, tgt, fused, lm_max_iterations=lm_max)
    tr = check_real(o, g, expect_trans=(1, 2) if lm_max <= 7 else (0, 2))
    acc = [r["accepted"] for r in tr if r["stage"] == 1]
    assert acc == ([1] + [0] * lm_max if lm_max <= 7 else [1] + [0] * 7 + [2])
    g.close()


@pytest.mark.parametrize("fused", [0, 1, 2])   # pass + controller launches / one launch per trial / one launch per frame (the resident kernel)
@pytest.mark.parametrize("max_it", [1, 2, 3])
def test_iteration_cap_on_real_clouds(clouds, fused, max_it):
    """max_iterations exhausted without convergence (:161 / :63): converged_ stays false, the last accepted pose comes back — both stages share the knob"""
    src, tgt = clouds
    o, g = make_real(src, tgt, fused, max_iterations=max_it, rotation_epsilon=1e-12, transformation_epsilon=1e-12)
    guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.03, -0.02, 0.06)
    check_real(o, g, guess, expect_rot=(0, max_it, False))
    g.close()


@pytest.mark.parametrize("fused", [0, 1, 2])   # pass + controller launches / one launch per trial / one launch per frame (the resident kernel)
@pytest.mark.parametrize("knobs", [dict(rotation_epsilon=1e-9, lm_init_lambda_factor=100.0), dict(rotation_epsilon=1e-4, transformation_epsilon=1e-5, lm_init_lambda_factor=1e-3),
                                   dict(lm_init_lambda_factor=1.0, lm_max_iterations=3), dict(rotation_epsilon=5e-2, transformation_epsilon=1e-1)])
def test_knobs_away_from_defaults_on_real_clouds(clouds, fused, knobs):
    """setRotationEpsilon / setTransformationEpsilon / setInitialLambdaFactor / lm_max_iterations with a 4 degree guess: exit, counts, pose and every significant trace record"""
    src, tgt = clouds
    o, g = make_real(src, tgt, fused, **knobs)
    guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.0, 0.0, math.radians(4.0))
    check_real(o, g, guess)
    g.close()


@pytest.mark.parametrize("fused", [0, 1, 2])   # pass + controller launches / one launch per trial / one launch per frame (the resident kernel)
def test_zero_budgets_on_real_clouds(clouds, fused):
    src, tgt = clouds
    o, g = make_real(src, tgt, fused, lm_max_iterations=0)
    guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.01, 0.0, 0.02)
    check_real(o, g, guess, expect_rot=(1, 1, False), expect_trans=(1, 1))
    assert np.array_equal(g.final_transformation_d, guess.astype(np.float64))
    g.close()
    o, g = make_real(src, tgt, fused, max_iterations=0)
    rc_o, _, Td_o, it_o, cv_o = o.align(guess)
    g.align(guess)
    assert (rc_o, it_o, cv_o) == (0, 0, False) and (g.last_stats.lm_failed, g.last_stats.n_outer, g.last_stats.converged) == (0, 0, 0)
    assert np.array_equal(g.final_transformation_d, Td_o)
    assert o.compute_translation(np.zeros(3), G, L0)[0] == -4   # no linearisation ever ran: no correspondences
    with pytest.raises(Exception, match="-4"):
        g.computeTranslation(np.zeros(3), G, L0)
    g.close()


@pytest.mark.parametrize("fused", [0, 1, 2])   # pass + controller launches / one launch per trial / one launch per frame (the resident kernel)
def test_setters_between_align_and_compute_translation_count(clouds, fused):
    """computeTranslation reads max_iterations_, lm_max_iterations_, transformation_epsilon_ and the lambda factor when IT runs (:63, :98, :142-148): a setter called
    after align() changes the translation stage — through round 5 the stage ran on the values the last align had copied into the device state"""
    src, tgt = clouds
    for knobs, expect in ((dict(lm_max_iterations=3), (1, 2)), (dict(max_iterations=1), (0, 1)), (dict(transformation_epsilon=1.0), (0, 1))):
        o, g = make_real(src, tgt, fused)
        assert o.align(None)[0] == 0
        g.align(None)
        o.set_driver_params(**knobs)
        for k, v in knobs.items():
            setattr(g._p, k, v)
        g._push()
        rc_t, t_o, tit_o = o.compute_translation(np.zeros(3), G, L0)
        t_g = g.computeTranslation(np.zeros(3), G, L0)
        ts = g.last_translation_stats
        assert (ts.lm_failed, ts.n_outer) == (rc_t, tit_o) == expect, (knobs, (ts.lm_failed, ts.n_outer), (rc_t, tit_o))
        assert np.abs(t_g - t_o).max() < 1e-8
        g.close()


def one_point_per_voxel(n_side=14):
    """a jittered lattice, one point per 1 m voxel (UNIFORM voxel k covers [k + 0.5, k + 1.5): the lattice sits on the integers), coordinates on a 2^-10 grid:
    source == target puts every source point exactly on its voxel's mean"""
    rng = np.random.default_rng(5)
    g = np.stack(np.meshgrid(np.arange(n_side), np.arange(n_side), np.arange(3), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    p = g + np.round(rng.uniform(-0.3, 0.3, g.shape) * 1024) / 1024 - np.array([n_side / 2, n_side / 2, 1.0])
    return np.ascontiguousarray(np.concatenate([p, np.ones((p.shape[0], 1))], 1).astype(np.float32))


@pytest.mark.parametrize("fused", [0, 1, 2])   # pass + controller launches / one launch per trial / one launch per frame (the resident kernel)
def test_nan_gain_ratio_on_a_crafted_scene(fused):
    """every residual is exactly zero => b = 0, d = 0, yi = y0 = 0: rho = 0 / 0 on both sides, `rho < 0` is false, the step is ACCEPTED (:307) and the stage ends
    converged after one iteration; the translation stage with ct_lambda = 0 does the same"""
    pts = one_point_per_voxel()
    kw = dict(voxel_type=1, voxel_resolution=1.0)
    o = pyorc.Reg(pyorc.default_params(**kw)); o.set_target(pts); o.set_source(pts.copy())
    g = RotVGICP(); g.setResolution(1.0); g._p.fused_lm = int(fused); g._push()
    g.setInputTarget(pts); g.setInputSource(pts.copy())
    rc_o, _, Td_o, it_o, cv_o = o.align(None)
    g.align(None)
    tr_o = [r for r in o.trace() if r["stage"] == 0]; tr_g = [r for r in g.trace() if r["stage"] == 0]
    assert (rc_o, it_o, cv_o) == (0, 1, True) and (g.last_stats.lm_failed, g.last_stats.n_outer, g.last_stats.converged) == (0, 1, 1)
    assert len(tr_o) == len(tr_g) == 1 and tr_o[0]["accepted"] == tr_g[0]["accepted"] == 1
    assert int(g.voxels()[1].max()) == 1                                  # one point per voxel
    assert math.isnan(tr_o[0]["rho"]) and math.isnan(tr_g[0]["rho"]) and tr_o[0]["y0"] == tr_g[0]["y0"] == 0.0 and tr_g[0]["yi"] == 0.0, (tr_o, tr_g)
    assert np.array_equal(g.final_transformation_d, np.eye(4)) and np.array_equal(Td_o, np.eye(4))
    z = np.zeros(3)
    rc_t, t_o, tit_o = o.compute_translation(z, z, z, ct_lambda=0.0)
    t_g = g.computeTranslation(z, z, z, ct_lambda=0.0)
    tt_o = [r for r in o.trace() if r["stage"] == 1]; tt_g = [r for r in g.trace() if r["stage"] == 1]
    assert (rc_t, tit_o) == (0, 1) and (g.last_translation_stats.lm_failed, g.last_translation_stats.n_outer) == (0, 1)
    assert len(tt_o) == len(tt_g) == 1 and math.isnan(tt_o[0]["rho"]) and math.isnan(tt_g[0]["rho"]) and tt_g[0]["accepted"] == 1
    assert np.array_equal(t_g, z) and np.array_equal(t_o, z)
    g.close()


def test_cpp_operator_prints_lm_not_converged(tmp_path, clouds):
    """the drop-in class writes the reference's diagnostic to stderr when a stage gives up (:66-69, :166-169) and still returns that stage's result: with three trials
    per iteration the translation stage of this pair fails in its second iteration (above), the rotation stage does not"""
    src, tgt = clouds
    exe = str(tmp_path / "shim_demo")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"), "-o", exe,
           "-L", os.path.join(ROOT, "rolo_amd"), "-lrolo_hip", "-Wl,-rpath," + os.path.join(ROOT, "rolo_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    src.astype(np.float32).tofile(tmp_path / "s.bin"); tgt.astype(np.float32).tofile(tmp_path / "t.bin")
    g3 = np.array([-0.28, -0.04, -0.02])
    for trials, fails in ((3, 1), (10, 0)):
        r = subprocess.run([exe, "lmfail", str(tmp_path / "s.bin"), str(tmp_path / "t.bin"), str(trials)], capture_output=True, text=True, check=True)
        assert r.stderr.count("lm not converged!!") == fails, r.stderr
        v = [float(x) for x in r.stdout.split()]
        o = pyorc.Reg(pyorc.default_params(polar_resolution=(0.175, 0.175, 2.0), voxel_type=0, lm_max_iterations=trials)); o.set_target(tgt); o.set_source(src)
        assert o.align(None)[0] == 0
        rc, t_o, it = o.compute_translation(np.zeros(3), g3, g3)
        assert rc == fails and np.abs(np.array(v[:3]) - t_o).max() < 1e-8 and int(v[4]) == src.shape[0]


_BAIL = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP
G = -np.asarray(synth.PREV_STEP_T); L0 = G * 0.97
src, tgt, _ = synth.dense_pair("vlp16", col_stride=2)
out = {}
for mode in (0, 2):
    g = RotVGICP(); g.setPolarResolution(0.175, 0.175, 2.0); g.setFusedLm(mode)
    res = []
    for k in range(4):   # eager, eager, captured graph, replay
        g.setInputTarget(tgt.copy()); g.setInputSource(src.copy())
        g.register_async(None, np.zeros(3), G, L0); Tf, Td, t = g.register_wait()
        res.append((Td.ravel().tolist(), np.asarray(t).tolist(), g.last_stats.n_outer, g.last_translation_stats.n_outer, g.last_stats.n_passes))
    g.align(None); Ta = g.final_transformation_d.ravel().tolist(); ta = g.computeTranslation(np.zeros(3), G, L0).tolist()
    out[str(mode)] = dict(res=res, align=Ta, trans=ta, counters=g.counters())
print(json.dumps(out))
"""


def test_resident_lm_kernel_gives_the_stage_back_when_it_cannot_become_resident():
    """fused_lm = 2 with nobody admitted (ROLO_LM_PERSIST_ADMIT_US=0, the test switch of the admission time-out): every launch of the resident kernel leaves WITHOUT touching the
    stage (LmState::lmp_bailed), the host finishes every frame with pass + controller launches — same poses as fused_lm = 0 to rounding, no error, the bails counted.
    This is the path two processes sharing a GPU (or more contexts in flight than fit) take instead of spinning on each other."""
    import json
    r = subprocess.run([sys.executable, "-c", _BAIL, ROOT], capture_output=True, text=True, env=dict(os.environ, ROLO_LM_PERSIST_ADMIT_US="0"), cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = out["0"], out["2"]
    assert b["counters"]["persist_bails"] >= 5 and a["counters"]["persist_bails"] == 0          # four frames + align (+ computeTranslation) all bailed
    for (Ta, ta, ia, ja, pa), (Tb, tb, ib, jb, pb) in zip(a["res"], b["res"]):
        assert (ia, ja, pa) == (ib, jb, pb)
        assert np.abs(np.array(Ta) - np.array(Tb)).max() < 1e-11 and np.abs(np.array(ta) - np.array(tb)).max() < 1e-11
    assert np.abs(np.array(a["align"]) - np.array(b["align"])).max() < 1e-11 and np.abs(np.array(a["trans"]) - np.array(b["trans"])).max() < 1e-11


# ---- the resident kernel's forms and its Mahalanobis cache -------------------------------------------------------------------------------------------
_FORMS = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP
sensor, leaf, hint = sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
src, tgt, _ = synth.dense_pair(sensor, seed=synth.SEED)
G = -np.asarray(synth.PREV_STEP_T, np.float64)
guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.01, -0.015, np.radians(3.0))
out = {}
for fused in (0, 2):
    rows = []
    for knobs in (dict(fixed_iterations=20), dict(), dict(lm_init_lambda_factor=1e-3, rotation_epsilon=1e-9), dict(lm_init_lambda_factor=100.0, lm_max_iterations=4)):
        g = RotVGICP(0); g.setResolution(leaf); g.setLoadHint(hint)
        for k, v in knobs.items():
            setattr(g._p, k, v)
        g._p.fused_lm = fused; g._push()
        g.setInputTarget(tgt); g.setInputSource(src)
        g.register_async(guess, np.zeros(3), G, G * 0.97)
        _, _, t = g.register_wait()
        st, ts = g.last_stats, g.last_translation_stats
        tr = [(r["stage"], r["outer"], r["trial"], r["accepted"], r["y0"], r["yi"]) for r in g.trace()]
        rows.append(dict(T=np.asarray(g.final_transformation_d).tolist(), t=np.asarray(t).tolist(), rot=[st.lm_failed, st.n_outer, int(st.converged), st.n_passes, st.n_cost_only, st.n_correspondences],
                         trans=[ts.lm_failed, ts.n_outer, ts.n_passes, ts.n_cost_only], trace=tr))
        g.close()
    out[str(fused)] = rows
print(json.dumps(out))
"""


@pytest.mark.parametrize("sensor,leaf,hint,wgs", [("os1-128", 0.5, 1, None), ("os1-64", 1.0, 1, None), ("os1-128", 0.5, 0, None), ("os1-128", 0.5, 1, "128")])
def test_resident_kernel_forms_and_cache_follow_the_pass_launches(sensor, leaf, hint, wgs):
    """The resident LM kernel in the forms the load picks — 64 workgroups x 4 points per thread (the headline's), x 2 (65 536-point clouds; 128 workgroups forced), 256 x 1 on an
    idle device — with its Mahalanobis cache (passes.hip lmp_rot_body: written by the (B) halves, read by the (A) halves and the translation passes, refilled after a rejected
    speculation) against pass + controller launches, which invert per trial: forced iterations (the cost-only run after convergence), the reference's own loop, a small initial
    damping (rejections early in the stage) and a large one with four trials per iteration (LM failure). Same exits, counts, cost-only passes and decisions; costs to 1e-9,
    poses to 1e-10 (the rows are sums over other groups of points)."""
    import json
    env = dict(os.environ)
    if wgs:
        env["ROLO_LM_PERSIST_WGS"] = wgs
    r = subprocess.run([sys.executable, "-c", _FORMS, ROOT, sensor, str(leaf), str(hint)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for a, b in zip(out["0"], out["2"]):
        # (cost-only passes, index 4 / 3: whether the pass after a trial is cost-only follows the trial's decision — where a stage goes on past convergence its decisions are
        # rounding noise of the sums, and the count with them; it is compared when no decision of the stage was noise)
        noisy = [any(r[0] == stage and abs(r[4] - r[5]) <= 1e-7 * abs(r[4]) for r in a["trace"]) for stage in (0, 1)]
        assert a["rot"][:4] + a["rot"][5:] == b["rot"][:4] + b["rot"][5:] and a["trans"][:3] == b["trans"][:3], (a["rot"], b["rot"], a["trans"], b["trans"])
        assert noisy[0] or a["rot"][4] == b["rot"][4], (a["rot"], b["rot"])
        assert noisy[1] or a["trans"][3] == b["trans"][3], (a["trans"], b["trans"])
        assert np.abs(np.array(a["T"]) - np.array(b["T"])).max() < 1e-10 and np.abs(np.array(a["t"]) - np.array(b["t"])).max() < 1e-10
        assert len(a["trace"]) == len(b["trace"])
        for ra, rb in zip(a["trace"], b["trace"]):
            if abs(ra[4] - ra[5]) <= 1e-7 * abs(ra[4]):      # a decision inside the rounding noise of the sums: the records after it may differ legitimately
                break
            assert ra[:4] == rb[:4], (ra, rb)
            assert same(ra[4], rb[4], 1e-9) and same(ra[5], rb[5], 1e-9), (ra, rb)
    # the scenarios do what they are there for: a cost-only run, rejections, an LM failure
    forced, own, small, large = out["2"]
    assert forced["rot"][1] == 20 and forced["rot"][4] >= 5
    assert any(r[3] == 0 for r in small["trace"]) or any(r[3] == 0 for r in own["trace"])


_BIG = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from rolo_amd import synth
from rolo_amd.rotvgicp import RotVGICP
src, tgt, _ = synth.dense_pair("os1-128x2048", seed=synth.SEED)          # 262 144 points per cloud
j = np.array([0.013, -0.007, 0.011, 0.0], np.float32)
src = np.concatenate([src, src + j]); tgt = np.concatenate([tgt, tgt + j])   # 524 288: four points per thread on 256 workgroups of the resident kernel
G = -np.asarray(synth.PREV_STEP_T, np.float64)
out = {}
for fused in (0, 2):
    g = RotVGICP(0); g.setResolution(0.5); g.setLoadHint(0); g._p.fused_lm = fused; g._push()
    g.setInputTarget(tgt); g.setInputSource(src)
    g.register_async(None, np.zeros(3), G, G * 0.97)
    _, _, t = g.register_wait()
    st, ts = g.last_stats, g.last_translation_stats
    out[str(fused)] = dict(T=np.asarray(g.final_transformation_d).tolist(), t=np.asarray(t).tolist(), rot=[st.lm_failed, st.n_outer, int(st.converged), st.n_correspondences],
                           trans=[ts.lm_failed, ts.n_outer], bails=g.counters()["persist_bails"], n=int(src.shape[0]))
    g.close()
print(json.dumps(out))
"""


def test_resident_kernel_on_a_cloud_beyond_the_cache_capacity():
    """524 288 points per cloud on an idle device: 256 workgroups x 4 points per thread — 60 KB of exchange rows + 96 KB of Mahalanobis cache would not fit a CU's 160 KB of LDS
    beside the state, so the launch takes the form without the cache (passes.hip launch_lm_persist: `fits`). Same exits and counts as pass + controller launches, poses to 1e-10,
    no bail-out."""
    import json
    r = subprocess.run([sys.executable, "-c", _BIG, ROOT], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = out["0"], out["2"]
    assert a["n"] == 524288 and b["bails"] == 0
    assert a["rot"] == b["rot"] and a["trans"] == b["trans"], (a, b)
    assert np.abs(np.array(a["T"]) - np.array(b["T"])).max() < 1e-10 and np.abs(np.array(a["t"]) - np.array(b["t"])).max() < 1e-10


@pytest.mark.parametrize("fused", [0, 2])
@pytest.mark.parametrize("reg", [1, 2, 4])   # MIN_EIG, NORMALIZED_MIN_EIG, FROBENIUS (rot_vgicp.hpp RegularizationMethod): no I - m m^T form — the passes read six entries per covariance
def test_solve_with_six_entry_covariances(clouds, fused, reg):
    """the regularisations whose covariance is not of the plane form: the passes (and the resident kernel's Mahalanobis cache, lmp_rotated_cov) rotate the six entries read
    from memory instead of I - m m^T — exits, counts, poses and the significant trace records against the oracle, a 3 degree guess"""
    src, tgt = clouds
    o, g = make_real(src, tgt, fused, regularization=reg)
    guess = np.eye(4, dtype=np.float32); guess[:3, :3] = synth.rpy_to_R(0.0, 0.01, math.radians(3.0))
    check_real(o, g, guess)
    g.close()
