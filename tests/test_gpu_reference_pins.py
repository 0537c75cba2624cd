"""-m gpu : the HIP path (through the C ABI) against fixtures made by RUNNING the reference's own function bodies
(tests/golden/make_golden.py round 3; CPU twin: tests/test_reference_pins.py).

  cost_golden / verdict_golden : scenario i with candidate-table row i is a reference rollout; the device rolls it,
      costs it (mpc.py:356-373) and judges it (mpc.py:296-321, 177-180) -- compared with what the reference's own
      CAV_utility and constraint builders said about the reference's trajectory.
  marshalling_golden           : MPC_Planner.update_initial_condition / update_predictions (mpc.py:241-294).
  augment_prev_sol_golden      : igtmpc.planner.augment_prev_sol with the device model object (utils.py:354-363)."""
import numpy as np
import pytest

import np_oracle as O
from helpers import REL_TOL, rel_err

pytestmark = pytest.mark.gpu
BITS = 0b11111


@pytest.fixture(scope='module')
def igt():
    import igtmpc
    igtmpc.load_library()
    return igtmpc


def _near_threshold(g, v, eps):
    """Golden rollouts with a judged quantity within eps of its threshold (the device's trajectory is ~1e-13 away from
    the reference's, so those verdicts are a coin toss at feas_tol = 0)."""
    X = g['X']
    near = (np.abs(np.abs(X[:, 3]) - 0.2) < eps).any(axis=1) | (np.abs(X[:, 5, :20] - 5) < eps).any(axis=1)
    near |= (np.abs(X[:, 5, :20]) < eps).any(axis=1)
    t = v['cinf_A'][:, 0] * X[:, 5, 19, None] + v['cinf_A'][:, 1] * g['U'][:, 0, 19, None] - v['cinf_b']
    return near | (np.abs(t) < eps).any(axis=1)


@pytest.mark.parametrize('dtype,tol,eps', [('f64', 1e-9, 1e-9), ('f32', REL_TOL, 2e-5)])
def test_device_cost_and_verdicts_equal_reference_functions(igt, golden_dir, dtype, tol, eps):
    g = np.load(f'{golden_dir}/frenet_rk4_golden.npz')
    v = np.load(f'{golden_dir}/verdict_golden.npz')
    c = np.load(f'{golden_dir}/cost_golden.npz')
    npdt = np.float64 if dtype == 'f64' else np.float32
    near = _near_threshold(g, v, eps)
    if dtype == 'f32':      # controls are rounded to float on entry: rate / box margins of the INPUTS move too
        U, up = g['U'], v['rolled_u_prev']
        da = np.abs(np.diff(np.concatenate([up[:, 0, None], U[:, 0]], axis=1), axis=1)) - 0.1 * 0.9
        dd = np.abs(np.diff(np.concatenate([up[:, 1, None], U[:, 1]], axis=1), axis=1)) - 0.1 * 0.7
        near |= (np.abs(da) < eps).any(axis=1) | (np.abs(dd) < eps).any(axis=1)
        P64 = O.Params(N=20)
        near |= O.breakpoint_distance(g['x0'].astype(npdt).astype(np.float64), U, g['kp'].astype(npdt).astype(np.float64),
                                      P64) < 1e-6
    print(f'{dtype}: rollouts set aside (a judged quantity within {eps:g} of its threshold): {near.sum()} of {len(near)}')
    assert near.mean() < (0.05 if dtype == 'f64' else 0.5)
    seen = set()
    with igt.BatchSolver(dtype=dtype, C=64, n_obs=0, cand_mode='table', feas_tol=0.0) as s:
        s.set_cinf(v['cinf_A'], v['cinf_b'])
        d = np.arange(64)
        for lo in range(0, 1024, 64):
            sl = slice(lo, lo + 64)
            s.set_candidate_table(g['U'][sl])
            out = s.rollout_all(g['x0'][sl].astype(npdt), v['rolled_u_prev'][sl].astype(npdt), g['kp'][sl].astype(npdt),
                                np.zeros(64, np.uint32), np.zeros((64, 0, 2, 21), npdt), want_X=False, want_U=False)
            ok = ~near[sl]
            assert rel_err(out['cost'][d, d][ok], c['rolled_J'][sl][ok]).max() <= tol
            got = (out['viol'][d, d] & BITS)[ok]
            want = v['rolled_bits'][sl][ok]
            assert np.array_equal(got, want), (lo, np.nonzero(got != want)[0][:5], got[got != want][:5], want[got != want][:5])
            seen |= set(want.tolist())
    assert len(seen) >= 10, 'the compared rows no longer mix the constraint families'


def test_device_judges_the_edge_controls_like_the_reference(igt, golden_dir):
    """The CONTROL-side edges of verdict_golden's hand-made cases (box a / df on the limit and one ulp outside, a rate
    step of exactly dt*jerk, the first step against u_prev) are inputs, so the device sees the very same numbers."""
    v = np.load(f'{golden_dir}/verdict_golden.npz')
    names = v['edge_names'].tolist()
    pick = [i for i, n in enumerate(names) if n.startswith(('a_', 'df_', 'rate_', 'all_inside'))]
    assert len(pick) >= 12
    U = np.zeros((64, 2, 20))
    U[:len(pick)] = v['edge_U'][pick]
    x0 = np.tile(np.array([10.0, 2.8, 10.0, 0.0, 0.0, 2.0, 0.0]), (len(pick), 1))
    kp = np.tile(np.array([np.inf, np.inf, 0.0]), (len(pick), 1))
    with igt.BatchSolver(dtype='f64', C=64, n_obs=0, cand_mode='table', feas_tol=0.0) as s:
        s.set_candidate_table(U)
        out = s.rollout_all(x0, v['edge_u_prev'][pick], kp, np.zeros(len(pick), np.uint32),
                            np.zeros((len(pick), 0, 2, 21)), want_X=False, want_U=False)
    d = np.arange(len(pick))
    got = out['viol'][d, d] & 0b110                 # box a/df and rate: the families decided by the controls alone
    want = v['edge_bits'][pick] & 0b110
    bad = [(names[pick[i]], bin(got[i]), bin(want[i])) for i in d if got[i] != want[i]]
    assert not bad, bad


def test_planner_marshalling_equals_reference(igt, golden_dir):
    m = np.load(f'{golden_dir}/marshalling_golden.npz')
    keys = ('x', 'y', 's', 'ey', 'epsi', 'v', 'heading')
    Ref, Act = igt.VehicleReference, igt.VehicleAction
    n_abs = 0
    for i in range(len(m['ind'])):
        routes = [str(r) for r in m['routes'][i]]
        ind = int(m['ind'][i])
        mk = lambda row: Ref(dict(zip(keys, row), K=None))
        agents = [{'type': 'CAV', 'state': mk(m['state'][i])} for _ in range(2)]
        pl = igt.MPC_Planner(N=20, dt=0.1, agents=agents, routes=routes, ref=None, index=ind, num_rk4_steps=4,
                             road_dim=(11.4, 50), ds_right=8.6, cand_mode='lattice')
        pl.update_initial_condition(agents[ind], Act({'a': m['u_prev'][i][0], 'df': m['u_prev'][i][1]}))
        flags = np.uint32(1 if pl._abs_heading[ind] else 0)
        assert np.array_equal(O.apply_flags(pl._x0[0], flags), m['x0_param'][i])      # what the kernel starts from
        assert np.array_equal(pl._u_prev[0], m['u_prev_param'][i])
        n_abs += int(flags) and m['state'][i][6] < 0
        preds = [[mk(m['preds'][i][a, k]) for k in range(21)] for a in range(2)]
        pl.update_predictions(preds, raw_preds=preds)
        assert np.array_equal(pl._obs[0, 0], m['preds_param'][i][:2])                   # rows x, y of the obstacle block
        assert np.array_equal(pl.raw_preds_np[0], m['raw_np'][i])
        j = 1 - ind
        assert np.array_equal(pl._tv_sv[0], m['raw_param'][i][[7 * j + 2, 7 * j + 5]])  # mpc.py:330
        assert pl.pred_ind == [j] and pl.NN_query_time == -1
    assert n_abs >= 4
    # ... and the kernel does apply |heading|: a '32' ego with heading -3 rolls from +3
    i = next(k for k in range(len(m['ind'])) if str(m['routes'][k][m['ind'][k]]) in ('32', '41') and m['state'][k][6] < 0)
    with igt.BatchSolver(dtype='f64', C=64, n_obs=0) as s:
        out = s.rollout_all(m['state'][i][None], np.zeros((1, 2)), np.array([[np.inf, np.inf, 0.0]]),
                            np.array([1], np.uint32), np.zeros((1, 0, 2, 21)), want_U=False)
    assert np.array_equal(out['X'][0, :, :, 0], np.tile(m['x0_param'][i], (64, 1)))


def test_augment_prev_sol_with_the_device_model_equals_reference(igt, golden_dir):
    from igtmpc.planner import augment_prev_sol
    a = np.load(f'{golden_dir}/augment_prev_sol_golden.npz')
    model = igt.KinematicBicycleModelFrenet(2.235, 2.235, 2.0, 0.1, discretization='rk4', mode='numpy', num_rk4_steps=4)
    worst = 0.0
    for i in range(len(a['kp'])):
        x, u = augment_prev_sol((a['x_sol_prev'][i], a['u_sol_prev'][i]), model, igt.Curvature(*a['kp'][i]))
        assert np.array_equal(u, a['u_aug'][i])
        assert np.array_equal(x[:, :-1], a['x_aug'][i][:, :-1])
        assert (x[5, -1] == 5) == (a['x_aug'][i][5, -1] == 5) and (x[5, -1] == -1) == (a['x_aug'][i][5, -1] == -1)
        worst = max(worst, rel_err(x[:, -1], a['x_aug'][i][:, -1]).max())
    assert worst < 1e-12, worst
