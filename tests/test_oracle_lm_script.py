"""CPU: the oracle's LM / GN drivers (oracle/rolo_oracle.cpp, restating lsq_registration_impl.hpp:55-179, 225-324) on SCRIPTED evaluations (orc_reg_set_script) —
the fixture of tests/test_gpu_lm_exits.py. Two things are held here, without a GPU: every scenario reaches the exit it is named after (EXPECT_* in
lm_scenarios.py), and the drivers agree with the independently written loops of the numpy twin (oracle/twin.py) fed the same script."""
import math
import warnings

import numpy as np
import pytest

from oracle import pyorc
from oracle.twin import Twin
from rolo_amd import synth
from lm_scenarios import make_script, ROT_SCENARIOS, TRANS_SCENARIOS, EXPECT_ROT, EXPECT_TRANS

G = -np.asarray(synth.PREV_STEP_T)
L0 = G * 0.97
GUESS = np.eye(4, dtype=np.float32); GUESS[:3, :3] = synth.rpy_to_R(0.01, -0.02, 0.07); GUESS[:3, 3] = (0.3, -0.1, 0.05)
START = np.array([0.01, -0.02, 0.005])


def oracle_rot(name, optimizer):
    params, outers = ROT_SCENARIOS[name]
    dof = 3 if optimizer == pyorc.OPT_SO3_LM else 6
    o = pyorc.Reg(pyorc.default_params(optimizer=optimizer, **params))
    o.set_script(*make_script(dof, outers))
    rc, _, Td, it, cv = o.align(GUESS)
    return o, rc, Td, it, cv


def oracle_trans(name):
    params, outers = TRANS_SCENARIOS[name]
    o = pyorc.Reg(pyorc.default_params())
    o.set_script(*make_script(3, ROT_SCENARIOS["iteration_cap_one"][1]))   # one rotation iteration leaves the correspondence count the stage divides by
    assert o.align(None)[0] == 0
    o.set_driver_params(**params); o.set_script(*make_script(6, outers, seed=11)); o.clear_trace()
    rc, t, it = o.compute_translation(START, G, L0)
    return o, rc, t, it


@pytest.mark.parametrize("optimizer", [pyorc.OPT_SO3_LM, pyorc.OPT_LM])
@pytest.mark.parametrize("name", sorted(ROT_SCENARIOS))
def test_rotation_scenarios_reach_their_exit(name, optimizer):
    o, rc, Td, it, cv = oracle_rot(name, optimizer)
    acc = [r["accepted"] for r in o.trace() if r["stage"] == 0]
    assert (rc, it, cv, acc) == EXPECT_ROT[name][3 if optimizer == pyorc.OPT_SO3_LM else 6]
    if rc == 1 and it == 1:   # "lm not converged!!" in the first iteration: the guess comes back untouched
        assert np.array_equal(Td, GUESS.astype(np.float64))
    tr = o.trace()
    if name == "nan_gain_ratio_is_accepted":
        assert math.isnan(tr[0]["rho"])
    if name == "minus_inf_gain_ratio":
        assert tr[0]["rho"] == -math.inf
    if name == "plus_inf_gain_ratio":
        assert tr[0]["rho"] == math.inf and abs(tr[0]["lam"]) > 0
    if name == "zero_gain_ratio_is_accepted":
        assert tr[0]["rho"] == 0.0 and tr[1]["lam"] == 2 * tr[0]["lam"]   # lambda * max(1/3, 1 - (-1)^3)
    if name == "damping_grows_until_converged":
        lam = [r["lam"] for r in tr]
        assert [lam[i + 1] / lam[i] for i in range(len(lam) - 1)] == [2.0, 4.0, 8.0][:len(lam) - 1]   # nu doubles with every rejection (:310-311)


@pytest.mark.parametrize("name", sorted(TRANS_SCENARIOS))
def test_translation_scenarios_reach_their_exit(name):
    o, rc, t, it = oracle_trans(name)
    acc = [r["accepted"] for r in o.trace() if r["stage"] == 1]
    assert (rc, it, acc) == EXPECT_TRANS[name]
    if rc == 1 and it == 1 or it == 0:
        assert np.array_equal(t, START)


class ScriptedTwin(Twin):
    """the twin's drivers (oracle/twin.py align / compute_translation — loops written independently of the C++ oracle's) on the same script"""

    def __init__(self, script, **kw):
        self.lin_y, self.lin_H, self.lin_b, self.lin_n, self.err = script
        self.max_iterations = kw.get("max_iterations", 64); self.rot_eps = kw.get("rotation_epsilon", 2e-3); self.trans_eps = kw.get("transformation_epsilon", 5e-4)
        self.lm_max = kw.get("lm_max_iterations", 10); self.lm_init = kw.get("lm_init_lambda_factor", 1e-9); self.fixed_iterations = kw.get("fixed_iterations", 0)
        self.q2_intended = 0; self.trace = []; self.o = -1; self.t = 0

    def _lin(self, dof):
        self.o += 1; self.t = 0
        o = min(self.o, len(self.lin_y) - 1)
        return self.lin_y[o], self.lin_H[o][:dof, :dof].copy(), self.lin_b[o][:dof].copy()

    def _err(self):
        o = min(self.o, len(self.lin_y) - 1); t = min(self.t, self.err.shape[1] - 1); self.t += 1
        return self.err[o][t]

    def so3_linearize(self, x0):
        return self._lin(3)

    def compute_error(self, xi):
        return self._err()

    def t3(self, t, g, l, dtn, dtn1, lam, error_variant, want_H=True):
        return self._err() if error_variant else self._lin(6)


@pytest.mark.parametrize("name", sorted(ROT_SCENARIOS))
def test_scripted_oracle_agrees_with_the_twins_drivers(name):
    params, outers = ROT_SCENARIOS[name]
    o, rc, Td, it, cv = oracle_rot(name, pyorc.OPT_SO3_LM)
    tw = ScriptedTwin(make_script(3, outers), **params)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        x0, it_t, cv_t, _ = tw.align(GUESS)
    assert (it_t, bool(cv_t)) == (it, cv)
    assert np.abs(x0 - Td).max() < 1e-12
    a = [r for r in o.trace() if r["stage"] == 0]
    assert [(r["outer"], r["trial"], r["accepted"]) for r in a] == [(r[1], r[2], r[3]) for r in tw.trace]
    for ro, rt in zip(a, tw.trace):
        for x, y in ((ro["y0"], rt[4]), (ro["yi"], rt[5]), (ro["rho"], rt[6]), (ro["lam"], rt[7])):
            assert (math.isnan(x) and math.isnan(y)) or x == y or abs(x - y) <= 1e-9 * max(abs(x), abs(y)), (name, ro, rt)
    # the twin's loop leaves without a flag when a step gives up: failure = the last iteration's trials are all rejections (or there was no trial to make)
    failed = tw.lm_max <= 0 or (len(tw.trace) > 0 and tw.trace[-1][3] == 0)
    assert int(failed) == rc or (tw.max_iterations <= 0 and rc == 0)


@pytest.mark.parametrize("name", sorted(TRANS_SCENARIOS))
def test_scripted_oracle_translation_agrees_with_the_twins_driver(name):
    params, outers = TRANS_SCENARIOS[name]
    o, rc, t, it = oracle_trans(name)
    tw = ScriptedTwin(make_script(6, outers, seed=11), **params)
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        t_t, it_t, _ = tw.compute_translation(START, G, L0)
    assert it_t == it and np.abs(t_t - t).max() < 1e-12
    a = [r for r in o.trace() if r["stage"] == 1]
    assert [(r["outer"], r["trial"], r["accepted"]) for r in a] == [(r[1], r[2], r[3]) for r in tw.trace]
    for ro, rt in zip(a, tw.trace):
        assert (math.isnan(ro["rho"]) and math.isnan(rt[6])) or abs(ro["rho"] - rt[6]) <= 1e-9 * abs(rt[6]) and abs(ro["lam"] - rt[7]) <= 1e-9 * abs(rt[7])
