"""Scripted evaluations for the LM / GN drivers' exit tests (tests/test_gpu_lm_exits.py on the GPU, tests/test_oracle_lm_script.py on the CPU):
the scenario tables, the script builder, and what each scenario is there to reach (EXPECT_*: the accept / reject pattern the oracle's drivers produce — a script
that drifts into another branch would still "match" between oracle and HIP path, so the pattern itself is pinned)."""
import numpy as np


def spd(dof, seed):
    """a fixed, well-conditioned normal matrix with off-diagonal entries (what sum J^T M J looks like: 1e6-ish)"""
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(6, 6))
    H = A @ A.T * 2e4 + np.diag([1e6, 2e6, 3e6, 4e6, 5e6, 6e6])
    if dof == 3:
        H[3:, :] = 0; H[:, 3:] = 0
    return H


def make_script(dof, outers, n_trial=12, pad=3, seed=7):
    """outers: list of dict(step=the Gauss-Newton step this linearisation asks for (dof values; b = -H step), y0, trials=[yi, ...], n=correspondences, same_as_prev)"""
    O = len(outers) + pad
    H = spd(dof, seed)
    lin_y = np.zeros(O); lin_H = np.zeros((O, 6, 6)); lin_b = np.zeros((O, 6)); lin_n = np.zeros(O, np.int32); err = np.zeros((O, n_trial))
    for o in range(O):
        d = outers[min(o, len(outers) - 1)]
        step = np.zeros(6); step[:dof] = np.asarray(d["step"], float)[:dof]
        lin_H[o] = H * d.get("hscale", 1.0)
        lin_b[o] = -(lin_H[o] @ step)
        lin_y[o] = d["y0"]; lin_n[o] = d.get("n", 5000)
        tr = list(d.get("trials", [d["y0"] * 0.5]))
        err[o] = [tr[min(t, len(tr) - 1)] for t in range(n_trial)]
    return lin_y, lin_H, lin_b, lin_n, err


BIG, MID, TINY = np.array([3e-2, -2e-2, 1e-2, 4e-2, -1e-2, 2e-2]), np.array([6e-3, 4e-3, -5e-3, 3e-3, -2e-3, 1e-3]), np.array([2e-6, -1e-6, 3e-6, 1e-6, 2e-6, -1e-6])

# name -> (parameter overrides, outers). y0 falls from outer to outer; a trial cost above y0 is a rejection (den = d^T (H + 2 lambda I) d > 0).
ROT_SCENARIOS = {
    "accept_until_converged": (dict(), [dict(step=BIG, y0=1e6, trials=[4e5]), dict(step=MID, y0=3.9e5, trials=[3.5e5]), dict(step=TINY, y0=3.4e5, trials=[3.39e5])]),
    "reject_twice_then_accept": (dict(), [dict(step=BIG, y0=1e6, trials=[1.2e6, 1.1e6, 6e5]), dict(step=TINY, y0=5.9e5, trials=[5.8e5])]),
    "lm_failure_in_second_outer": (dict(lm_max_iterations=3), [dict(step=BIG, y0=1e6, trials=[4e5]), dict(step=MID, y0=3.9e5, trials=[4.5e5, 4.4e5, 4.3e5, 1e5])]),
    "lm_failure_first_trial": (dict(lm_max_iterations=1), [dict(step=BIG, y0=1e6, trials=[1.5e6, 1e5])]),
    "lm_failure_two_trials": (dict(lm_max_iterations=2), [dict(step=BIG, y0=1e6, trials=[1.5e6, 1.4e6, 1e5])]),
    "rejected_but_converged": (dict(), [dict(step=BIG, y0=1e6, trials=[4e5]), dict(step=TINY, y0=3.9e5, trials=[3.95e5])]),
    "damping_grows_until_converged": (dict(lm_init_lambda_factor=1.0, lm_max_iterations=12), [dict(step=BIG, y0=1e6, trials=[1.1e6])]),   # lambda x2 x4 x8 ... shrinks the step below epsilon
    "damping_grows_until_lm_failure": (dict(lm_init_lambda_factor=1.0, lm_max_iterations=2), [dict(step=BIG, y0=1e6, trials=[1.1e6])]),
    "iteration_cap": (dict(max_iterations=2), [dict(step=BIG, y0=1e6, trials=[4e5]), dict(step=BIG, y0=3.9e5, trials=[3e5]), dict(step=BIG, y0=2.9e5, trials=[2e5])]),
    "iteration_cap_one": (dict(max_iterations=1), [dict(step=BIG, y0=1e6, trials=[4e5])]),
    "tight_epsilons": (dict(rotation_epsilon=1e-9, transformation_epsilon=1e-9, max_iterations=5), [dict(step=TINY, y0=1e6, trials=[9e5])]),
    "loose_epsilons": (dict(rotation_epsilon=1e-1, transformation_epsilon=1e-1), [dict(step=BIG, y0=1e6, trials=[9e5])]),
    "nan_gain_ratio_is_accepted": (dict(), [dict(step=0 * BIG, y0=1e6, trials=[1e6])]),            # b = 0 => d = 0 => rho = 0 / 0: `rho < 0` is false (:307)
    "minus_inf_gain_ratio": (dict(), [dict(step=0 * BIG, y0=1e6, trials=[2e6])]),                  # (y0 - yi) / 0 = -inf: rejected, delta = I => converged
    "plus_inf_gain_ratio": (dict(), [dict(step=0 * BIG, y0=1e6, trials=[5e5])]),                   # +inf: accepted, lambda * max(1/3, 1 - inf) = lambda / 3
    "zero_gain_ratio_is_accepted": (dict(), [dict(step=BIG, y0=1e6, trials=[1e6]), dict(step=TINY, y0=1e6, trials=[9e5])]),   # rho = +0: accepted, lambda doubles
    "no_trial_budget": (dict(lm_max_iterations=0), [dict(step=BIG, y0=1e6, trials=[4e5])]),
    "no_iteration_budget": (dict(max_iterations=0), [dict(step=BIG, y0=1e6, trials=[4e5])]),
    "forced_iterations_repeat_a_converged_rejection": (dict(fixed_iterations=6), [dict(step=BIG, y0=1e6, trials=[4e5]), dict(step=TINY, y0=3.9e5, trials=[3.95e5]),
                                                                              dict(step=TINY, y0=3.9e5, trials=[3.95e5])]),
    "forced_iterations_accepting": (dict(fixed_iterations=4), [dict(step=BIG, y0=1e6, trials=[4e5]), dict(step=MID, y0=3.9e5, trials=[4e5, 3.8e5]), dict(step=TINY, y0=3.7e5, trials=[3.6e5])]),
}



# the translation stage: only delta's translation decides convergence (is_t_converged, :142-148)
TBIG, TMID, TTINY = np.array([0, 0, 0, 4e-2, -1e-2, 2e-2]), np.array([1e-3, 0, 0, 3e-3, -2e-3, 1e-3]), np.array([0, 1e-3, 0, 1e-6, 2e-6, -1e-6])
TRANS_SCENARIOS = {
    "accept_until_converged": (dict(), [dict(step=TBIG, y0=3e6, trials=[2.5e6]), dict(step=TMID, y0=2.4e6, trials=[2.3e6]), dict(step=TTINY, y0=2.2e6, trials=[2.1e6])]),
    "rejections_until_converged": (dict(lm_init_lambda_factor=2.0, lm_max_iterations=12), [dict(step=TBIG, y0=3e6, trials=[2.5e6]), dict(step=TMID, y0=2.4e6, trials=[2.5e6])]),
    "lm_failure": (dict(lm_max_iterations=4), [dict(step=TBIG, y0=3e6, trials=[2.5e6]), dict(step=TMID, y0=2.4e6, trials=[2.5e6, 2.6e6, 2.7e6, 2.8e6, 1e5])]),
    "lm_failure_first_trial": (dict(lm_max_iterations=1), [dict(step=TBIG, y0=3e6, trials=[3.5e6])]),
    "rejected_but_converged": (dict(), [dict(step=TBIG, y0=3e6, trials=[2.5e6]), dict(step=TTINY, y0=2.4e6, trials=[2.45e6])]),
    "reject_then_accept": (dict(), [dict(step=TBIG, y0=3e6, trials=[3.1e6, 3.2e6, 2.5e6]), dict(step=TTINY, y0=2.4e6, trials=[2.3e6])]),
    "iteration_cap": (dict(max_iterations=2), [dict(step=TBIG, y0=3e6, trials=[2.5e6]), dict(step=TBIG, y0=2.4e6, trials=[2.3e6]), dict(step=TBIG, y0=2.2e6, trials=[2.1e6])]),
    "tight_epsilon": (dict(transformation_epsilon=1e-9, max_iterations=4), [dict(step=TTINY, y0=3e6, trials=[2.9e6])]),
    "nan_gain_ratio_is_accepted": (dict(), [dict(step=0 * TBIG, y0=3e6, trials=[3e6])]),
    "no_trial_budget": (dict(lm_max_iterations=0), [dict(step=TBIG, y0=3e6, trials=[2.5e6])]),
    "no_iteration_budget": (dict(max_iterations=0), [dict(step=TBIG, y0=3e6, trials=[2.5e6])]),
}




# {degrees of freedom: (lm_failed, n_outer, converged, accepted-flags of the trace)} of every rotation scenario; (lm_failed, n_outer, flags) of the translation ones.
# 1 accepted, 0 rejected, 2 rejected-but-converged. Written down once from the oracle's run and read against each scenario's intent; the CPU test holds the oracle to it.
EXPECT_ROT = {
    'accept_until_converged': {3: (0, 3, True, [1, 1, 1]), 6: (0, 3, True, [1, 1, 1])},
    'damping_grows_until_converged': {3: (0, 1, True, [0, 0, 2]), 6: (0, 1, True, [0, 0, 0, 2])},
    'damping_grows_until_lm_failure': {3: (1, 1, False, [0, 0]), 6: (1, 1, False, [0, 0])},
    'forced_iterations_accepting': {3: (0, 4, True, [1, 0, 1, 1, 1]), 6: (0, 4, True, [1, 0, 1, 1, 1])},
    'forced_iterations_repeat_a_converged_rejection': {3: (0, 6, True, [1, 2, 2, 2, 2, 2]), 6: (0, 6, True, [1, 2, 2, 2, 2, 2])},
    'iteration_cap': {3: (0, 2, False, [1, 1]), 6: (0, 2, False, [1, 1])},
    'iteration_cap_one': {3: (0, 1, False, [1]), 6: (0, 1, False, [1])},
    'lm_failure_first_trial': {3: (1, 1, False, [0]), 6: (1, 1, False, [0])},
    'lm_failure_in_second_outer': {3: (1, 2, False, [1, 0, 0, 0]), 6: (1, 2, False, [1, 0, 0, 0])},
    'lm_failure_two_trials': {3: (1, 1, False, [0, 0]), 6: (1, 1, False, [0, 0])},
    'loose_epsilons': {3: (0, 1, True, [1]), 6: (0, 1, True, [1])},
    'minus_inf_gain_ratio': {3: (0, 1, True, [2]), 6: (0, 1, True, [2])},
    'nan_gain_ratio_is_accepted': {3: (0, 1, True, [1]), 6: (0, 1, True, [1])},
    'no_iteration_budget': {3: (0, 0, False, []), 6: (0, 0, False, [])},
    'no_trial_budget': {3: (1, 1, False, []), 6: (1, 1, False, [])},
    'plus_inf_gain_ratio': {3: (0, 1, True, [1]), 6: (0, 1, True, [1])},
    'reject_twice_then_accept': {3: (0, 2, True, [0, 0, 1, 1]), 6: (0, 2, True, [0, 0, 1, 1])},
    'rejected_but_converged': {3: (0, 2, True, [1, 2]), 6: (0, 2, True, [1, 2])},
    'tight_epsilons': {3: (0, 5, False, [1, 1, 1, 1, 1]), 6: (0, 5, False, [1, 1, 1, 1, 1])},
    'zero_gain_ratio_is_accepted': {3: (0, 2, True, [1, 1]), 6: (0, 2, True, [1, 1])},
}
EXPECT_TRANS = {
    'accept_until_converged': (0, 3, [1, 1, 1]),
    'iteration_cap': (0, 2, [1, 1]),
    'lm_failure': (1, 2, [1, 0, 0, 0, 0]),
    'lm_failure_first_trial': (1, 1, [0]),
    'nan_gain_ratio_is_accepted': (0, 1, [1]),
    'no_iteration_budget': (0, 0, []),
    'no_trial_budget': (1, 1, []),
    'reject_then_accept': (0, 2, [0, 0, 1, 1]),
    'rejected_but_converged': (0, 2, [1, 2]),
    'rejections_until_converged': (0, 2, [1, 0, 0, 2]),
    'tight_epsilon': (0, 4, [1, 1, 1, 1]),
}
