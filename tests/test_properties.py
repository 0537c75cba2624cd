"""CPU: property tests (hypothesis; SURVEY section 4's test plan) of the oracle and the host logic -- invariances the domain
offers, checked on drawn inputs rather than on fixed seeds."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import np_oracle as O

SET = settings(max_examples=12, deadline=None, suppress_health_check=[HealthCheck.too_slow])


def _batch(seed, B):
    from igtmpc.scenarios import make_batch
    return make_batch(B, dtype=np.float64, seed=seed)


@SET
@given(seed=st.integers(0, 10 ** 6), perm_seed=st.integers(0, 10 ** 6))
def test_solve_is_invariant_under_permutation_of_the_batch(seed, perm_seed):
    """Scenarios are independent (evaluate.py:469-558): permuting the batch permutes the answers, bit for bit."""
    b = _batch(seed, 6)
    P = O.Params(N=10)
    f = lambda k, idx: b[k][idx][..., :P.N + 1] if k == 'obs_xy' else b[k][idx]
    idx = np.arange(6)
    perm = np.random.default_rng(perm_seed).permutation(6)
    r0 = O.solve_batch(*[f(k, idx) for k in ('x0', 'u_prev', 'kparams', 'flags', 'obs_xy')], None, None, P, C=64)
    r1 = O.solve_batch(*[f(k, perm) for k in ('x0', 'u_prev', 'kparams', 'flags', 'obs_xy')], None, None, P, C=64)
    for k in ('x', 'u', 'cost', 'argmin', 'status'):
        assert np.array_equal(r0[k][perm], r1[k], equal_nan=True), k


@SET
@given(seed=st.integers(0, 10 ** 6), order_seed=st.integers(0, 10 ** 6))
def test_argmin_does_not_depend_on_the_order_of_the_candidates(seed, order_seed):
    """Ties go to the lowest candidate index (the device: (J, c) lexicographic): presenting the same table in another
    order gives the same control sequence and cost, and the index of that sequence's first occurrence."""
    b = _batch(seed, 3)
    P = O.Params(N=8)
    U = O.candidates_lattice(b['u_prev'][:1], P, 64)[0]
    U = np.concatenate([U, U[:16]])                              # exact duplicates: ties
    order = np.random.default_rng(order_seed).permutation(len(U))
    args = (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'][..., :P.N + 1], None, None, P)
    r0 = O.solve_batch(*args, C=len(U), U=U)
    r1 = O.solve_batch(*args, C=len(U), U=U[order])
    assert np.array_equal(r0['status'], r1['status']) and np.array_equal(r0['cost'], r1['cost'], equal_nan=True)
    assert np.array_equal(r0['u'], r1['u'], equal_nan=True)
    for i in range(3):
        if r1['status'][i] == 0:
            same = (U[order] == r1['u'][i]).all(axis=(1, 2))
            assert r1['argmin'][i] == np.nonzero(same)[0].min()


@SET
@given(x=st.floats(-30, 60), y=st.floats(-30, 60), h=st.floats(-3.2, 3.2), ox=st.floats(-30, 60), oy=st.floats(-30, 60),
       shift=st.floats(-20, 20))
def test_filter_preds_depends_on_the_relative_pose_only(x, y, h, ox, oy, shift):
    """utils.py:365-388: the verdict is the sign of (p_obs - p_ego) . (cos psi, sin psi): translating both vehicles by the
    same vector changes nothing, and an obstacle mirrored through the ego lands on the other side."""
    from igtmpc import routes as R
    obs = np.zeros((1, 1, 2, 5)); obs[0, 0, 0] = ox; obs[0, 0, 1] = oy
    a = R.filter_preds(np.array([[x, y]]), np.array([h]), obs)
    sh = obs + shift
    b2 = R.filter_preds(np.array([[x + shift, y + shift]]), np.array([h]), sh)
    dot = (ox - x) * np.cos(h) + (oy - y) * np.sin(h)
    if abs(dot) > 1e-6:                                           # away from the boundary the shifted scene decides alike
        assert (a[0, 0, 0, 0] == -20.0) == (b2[0, 0, 0, 0] == -20.0)
        mirrored = np.zeros_like(obs); mirrored[0, 0, 0] = 2 * x - ox; mirrored[0, 0, 1] = 2 * y - oy
        c = R.filter_preds(np.array([[x, y]]), np.array([h]), mirrored)
        assert (a[0, 0, 0, 0] == -20.0) != (c[0, 0, 0, 0] == -20.0)
        assert (a[0, 0, 0, 0] == -20.0) == (dot < 0)


@SET
@given(v=st.floats(-0.9, 4.9), a=st.floats(-3.9, 2.9), steps=st.integers(1, 30), seed=st.integers(0, 10 ** 6))
def test_cinf_is_control_invariant(v, a, steps, seed):
    """mpc.py:88-104: from any (v, a) inside C_inf there is a jerk-feasible sequence that stays inside forever -- taking,
    at every step, the admissible jerk that keeps the successor inside (it exists by construction) never leaves the box."""
    from igtmpc.cinf import cinf_halfplanes
    A, b = cinf_halfplanes(dt=0.1, jerk=0.9)
    inside = lambda z: (A @ z - b).max() <= 1e-9
    z = np.array([v, a])
    if not inside(z):
        return
    rng = np.random.default_rng(seed)
    for _ in range(steps):
        nxt = None
        for da in np.concatenate([rng.permutation(np.linspace(-0.09, 0.09, 19))]):
            cand = np.array([z[0] + 0.1 * z[1], z[1] + da])      # A = [[1, dt], [0, 1]], B = [0, 1]
            if inside(cand):
                nxt = cand
                break
        assert nxt is not None, z
        z = nxt
        assert -1 - 1e-9 <= z[0] <= 5 + 1e-9 and -4 - 1e-9 <= z[1] <= 3 + 1e-9
