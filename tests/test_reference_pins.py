"""CPU: the oracle and the host logic against fixtures made by RUNNING the reference's own function bodies
(tests/golden/make_golden.py, round 3: FunctionDef nodes of mpc.py / common/utils.py compiled unmodified and fed numpy).

What each fixture pins (reference file:line):
  cost_golden.npz              mpc.py:356-373  CAV_utility
  verdict_golden.npz           mpc.py:296-321 box / rate / |ey| builders, 177-180 terminal index convention
  scenario_encoding.json       utils.py:84-169, 393-402
  filter_preds_golden.npz      utils.py:365-388
  augment_prev_sol_golden.npz  utils.py:354-363 (with the reference's numpy RK4 model)
  marshalling_golden.npz       mpc.py:241-294 (set_value sink)
  value_features_golden.npz    mpc.py:326-354 get_xN(mode='numpy')
  frenet2global_straight_golden.npz  utils.py:532-553
The HIP path against the same fixtures: tests/test_gpu_reference_pins.py."""
import json
import types

import numpy as np
import pytest

import closed_loop as CL
import np_oracle as O

BITS = 0b11111        # box v, box a/df, rate, |ey|, terminal set -- the families the fixtures cover


def _z(golden_dir, name):
    return np.load(f'{golden_dir}/{name}')


def test_oracle_cost_equals_reference_cav_utility(golden_dir):
    c = _z(golden_dir, 'cost_golden.npz')
    g = _z(golden_dir, 'frenet_rk4_golden.npz')
    J = O.stage_cost(g['X'], g['U'], O.Params(N=20))
    assert np.array_equal(J, c['rolled_J'])                       # 1024 real trajectories, 0 ulp
    for N in (20, 40, 10):
        J = O.stage_cost(c[f'random_N{N}_X'], c[f'random_N{N}_U'], O.Params(N=N))
        assert np.array_equal(J, c[f'random_N{N}_J'])


def test_planner_cav_utility_equals_reference(golden_dir):
    """The product's host-side MPC_Planner.CAV_utility (igtmpc/planner.py) on the same arrays."""
    from igtmpc.planner import MPC_Planner
    c = _z(golden_dir, 'cost_golden.npz')
    for N in (20, 40, 10):
        me = types.SimpleNamespace(N=N)
        J = np.array([MPC_Planner.CAV_utility(me, x, u) for x, u in zip(c[f'random_N{N}_X'], c[f'random_N{N}_U'])])
        assert np.array_equal(J, c[f'random_N{N}_J'])


def test_oracle_verdicts_equal_reference_constraint_builders(golden_dir):
    v = _z(golden_dir, 'verdict_golden.npz')
    g = _z(golden_dir, 'frenet_rk4_golden.npz')
    P = O.Params(N=20, feas_tol=0.0)                               # the builders are exact inequalities
    A, b = v['cinf_A'], v['cinf_b']
    _, mask = O.constraint_violation(g['X'], g['U'], v['rolled_u_prev'], None, A, b, P)
    assert np.array_equal(mask & BITS, v['rolled_bits'])
    assert len(set(v['rolled_bits'].tolist())) >= 12               # the sample mixes the families
    _, mask = O.constraint_violation(v['edge_X'], v['edge_U'], v['edge_u_prev'], None, A, b, P)
    bad = [(n, bin(w), bin(m)) for n, w, m in zip(v['edge_names'], v['edge_bits'], mask & BITS) if w != m]
    assert not bad, bad
    # the k-range edges really are in the fixture (and say what SURVEY section 7 hard part 4 says)
    bits = dict(zip(v['edge_names'].tolist(), v['edge_bits'].tolist()))
    assert bits['vN_above_5_accepted'] == 0 and bits['eyN_rejected'] == 8
    assert bits['terminal_outside_at_Nm1'] == 16 and bits['terminal_pair_at_Nm2_accepted'] == 0
    assert bits['terminal_ignores_xN'] == 0 and bits['v_exactly_5'] == 0 and bits['v_one_ulp_above_5'] == 1
    assert bits['rate_first_step_exactly_dt_jerk_accepted'] == 0 and bits['rate_first_step_vs_u_prev'] == 4


def test_c_oracle_cost_and_verdicts_equal_reference(golden_dir):
    """The plain-C restatement (the timed CPU baseline): scenario i with table row i is the golden pair; cost at 1e-12,
    verdict bits exact where no judged quantity sits within 1e-9 of its threshold (the C rollout is ~1e-15 off the
    reference's)."""
    import c_oracle as CO
    v = _z(golden_dir, 'verdict_golden.npz')
    g = _z(golden_dir, 'frenet_rk4_golden.npz')
    c = _z(golden_dir, 'cost_golden.npz')
    P = O.Params(N=20, feas_tol=0.0)
    X = g['X']
    near = (np.abs(np.abs(X[:, 3]) - 0.2) < 1e-9).any(axis=1) | (np.abs(X[:, 5, :20] - 5) < 1e-9).any(axis=1)
    near |= (np.abs(X[:, 5, :20]) < 1e-9).any(axis=1)
    t = v['cinf_A'][:, 0] * X[:, 5, 19, None] + v['cinf_A'][:, 1] * g['U'][:, 0, 19, None] - v['cinf_b']
    near |= (np.abs(t) < 1e-9).any(axis=1)
    assert near.mean() < 0.05
    for lo in range(0, 1024, 64):
        sl = slice(lo, lo + 64)
        out = CO.rollout_all(g['x0'][sl], v['rolled_u_prev'][sl], g['kp'][sl], np.zeros(64, np.uint32), None,
                             v['cinf_A'], v['cinf_b'], P, C=64, U=g['U'][sl])
        d = np.arange(64)
        assert np.abs(out['cost'][d, d] - c['rolled_J'][sl]).max() < 1e-12
        # the rate family is judged against each SCENARIO's u_prev; row i of the table belongs to scenario i
        ok = ~near[sl]
        assert np.array_equal((out['viol'][d, d] & BITS)[ok], v['rolled_bits'][sl][ok])


def test_scenario_and_route_encoding_equal_reference(golden_dir):
    from igtmpc import routes as R
    with open(f'{golden_dir}/scenario_encoding.json') as f:
        fx = json.load(f)
    n_ok = 0
    for key, want in fx['scenario_encoding'].items():
        pair = key.split(',')
        if want is None:
            with pytest.raises(ValueError):
                R.scenario_of(pair)
        else:
            assert list(R.scenario_encoding_sign(pair, R.scenario_of(pair))) == want, key
            n_ok += 1
    assert n_ok == 60            # 7 scenarios x 4 sets x 2 orders + scenario 7's 2 distinct sets x 2
    assert fx['route_encoding'] == R.ROUTE_ID


def test_filter_preds_equals_reference(golden_dir):
    from igtmpc import routes as R
    f = _z(golden_dir, 'filter_preds_golden.npz')
    got = R.filter_preds(f['ego_xyh'][:, :2], f['ego_xyh'][:, 2], f['obs_xy'][:, None])[:, 0]
    assert np.array_equal(got, f['filtered_xy'])
    moved = (f['filtered_xy'] != f['obs_xy']).any(axis=(1, 2))
    assert 0.3 < moved.mean() < 0.7
    for i in range(len(got)):                                       # the oracle's scalar form
        x, y = O.filter_preds_xy(f['ego_xyh'][i, :2], f['ego_xyh'][i, 2], f['obs_xy'][i, 0], f['obs_xy'][i, 1])
        assert np.array_equal(np.stack([x, y]), f['filtered_xy'][i])


def test_augment_prev_sol_equals_reference(golden_dir):
    a = _z(golden_dir, 'augment_prev_sol_golden.npz')
    P = O.Params(N=20)
    hit5 = hitm1 = 0
    for i in range(len(a['kp'])):
        x, u = CL.augment_prev_sol(a['x_sol_prev'][i], a['u_sol_prev'][i], a['kp'][i], P)
        assert np.array_equal(x, a['x_aug'][i]) and np.array_equal(u, a['u_aug'][i])
        hit5 += a['x_aug'][i, 5, -1] == 5
        hitm1 += a['x_aug'][i, 5, -1] == -1
    assert hit5 > 10 and hitm1 > 10
    # the product's host function (igtmpc/planner.py) with a model object built on the oracle's step
    from igtmpc.planner import augment_prev_sol
    from igtmpc.vehicle import Curvature, VehicleReference

    def model(state, action):
        nxt = O.frenet_rk4_step(np.array([state.x, state.y, state.s, state.ey, state.epsi, state.v, state.heading]),
                                action.a, action.df, np.array(state.K.kparams), P)
        return VehicleReference(dict(zip(('x', 'y', 's', 'ey', 'epsi', 'v', 'heading'), nxt), K=state.K))
    for i in range(0, len(a['kp']), 3):
        x, u = augment_prev_sol((a['x_sol_prev'][i], a['u_sol_prev'][i]), model, Curvature(*a['kp'][i]))
        assert np.array_equal(x, a['x_aug'][i]) and np.array_equal(u, a['u_aug'][i])


def test_marshalling_flags_equal_reference(golden_dir):
    """mpc.py:280-294: x0[6] = |heading| exactly on the routes '32' and '41' -- what flag bit 0 stands for."""
    from igtmpc import routes as R
    m = _z(golden_dir, 'marshalling_golden.npz')
    n_abs = 0
    for i in range(len(m['ind'])):
        route = str(m['routes'][i][m['ind'][i]])
        flag = np.uint32(1 if R.TABLES['abs_heading'][R.ROUTE_ID[route]] else 0)
        assert np.array_equal(O.apply_flags(m['state'][i], flag), m['x0_param'][i])
        assert np.array_equal(m['u_prev_param'][i], m['u_prev'][i])
        n_abs += int(flag) and m['state'][i][6] < 0
        # obstacle block: rows x, y, s, ey, epsi, v as they come; heading by the OBSTACLE's route (mpc.py:250-253)
        j = 1 - m['ind'][i]
        blk = m['preds'][i][j].T.copy()
        if str(m['routes'][i][j]) in ('32', '41'):
            blk[6] = np.abs(blk[6])
        assert np.array_equal(blk, m['preds_param'][i])
        raw = m['preds'][i][:, -1, :].copy()
        assert np.array_equal(raw.reshape(-1), m['raw_np'][i])      # raw_preds_np keeps the signed heading (mpc.py:272)
        for a in range(2):
            if str(m['routes'][i][a]) in ('32', '41'):
                raw[a, 6] = abs(raw[a, 6])
        assert np.array_equal(raw.reshape(-1), m['raw_param'][i])
    assert n_abs >= 4


def test_value_features_equal_reference(golden_dir):
    from igtmpc import routes as R
    f = _z(golden_dir, 'value_features_golden.npz')
    for i in range(len(f['ind'])):
        pair = [str(r) for r in f['routes'][i]]
        ind = int(f['ind'][i])
        j = 1 - ind
        e = R.scenario_encoding_sign(pair, R.scenario_of(pair))
        tv = f['raw_np'][i][[7 * j + 2, 7 * j + 5]]                 # mpc.py:330: the other agent's last raw prediction
        got = O.value_features(f['sN_vN'][i][None, None, 0], f['sN_vN'][i][None, None, 1], tv[None],
                               np.array([[e[ind], e[j]]], dtype=np.float64))[0, 0]
        assert np.array_equal(got, f['xN_scenario'][i])
        r = [R.ROUTE_ID[p] for p in pair]
        got = O.value_features(f['sN_vN'][i][None, None, 0], f['sN_vN'][i][None, None, 1], tv[None],
                               np.array([[r[ind], r[j]]], dtype=np.float64))[0, 0]
        assert np.array_equal(got, f['xN_route'][i])


def test_frenet2global_straight_routes_equal_reference(golden_dir):
    from igtmpc import routes as R
    f = _z(golden_dir, 'frenet2global_straight_golden.npz')
    consts = CL.route_constants()
    for route in ('13', '24', '31', '42'):
        ref = np.array([O.frenet2global_ref(route, s, consts[route]) for s in f['s']])
        assert np.array_equal(ref, f[route])
        got = R.frenet2global(np.full(len(f['s']), R.ROUTE_ID[route]), f['s'])
        assert np.abs(got - f[route]).max() < 1e-12
