"""CPU: the oracle (numpy + C restatements) against the reference-generated golden vectors
(tests/golden/*.npz, made by tests/golden/make_golden.py) and the SURVEY section 4 KAT."""
import json

import numpy as np
import pytest

import np_oracle as O


def test_frenet_rollout_bit_exact(golden_dir):
    g = np.load(f'{golden_dir}/frenet_rk4_golden.npz')
    X = O.rollout_frenet(g['x0'], g['U'], g['kp'], O.Params(n_rk4=int(g['n_rk4']), dt=float(g['dt'])))
    assert np.array_equal(X, g['X'])          # 1024 rollouts x 20 steps, 0 ulp


def test_frenet_rollout_rk7_bit_exact(golden_dir):
    g = np.load(f'{golden_dir}/frenet_rk4_golden_rk7.npz')
    X = O.rollout_frenet(g['x0'], g['U'], g['kp'], O.Params(n_rk4=7))
    assert np.array_equal(X, g['X'])


def test_cartesian_euler_bit_exact(golden_dir):
    g = np.load(f'{golden_dir}/cartesian_euler_golden.npz')
    z = O.cartesian_euler_step(g['z'], g['u'][:, 0], g['u'][:, 1], O.Params())
    assert np.array_equal(z, g['z_next'])


def test_survey_known_answer_vectors():
    """SURVEY.md section 4 table (recorded from the reference during the survey)."""
    k = np.arange(20)
    U = np.stack([0.5 * np.cos(0.3 * k), 0.1 * np.sin(0.2 * k)])
    x0 = np.array([18, 2.8, 18, 0.05, -0.02, 3, 0.0])
    kp = np.array([19.3, 19.3 + 8.6 * np.pi / 2, 1 / 8.6])
    X = O.rollout_frenet(x0, U, kp, O.Params())
    want = {1: [18.302499999999995, 2.8, 18.30243950201664, 0.04395040332526674, -0.02, 3.0500000000000007, 0.0],
            10: [21.127355623760682, 2.956599698035115, 21.132911009599095, -0.05072384947021227,
                 -0.18613145838029138, 3.0730931485799537, 0.046632589278592346],
            20: [24.04054200868725, 3.1707601439799964, 23.768492505124048, -0.9596488518839958,
                 -0.47896770505284986, 2.9547762875359105, 0.06025930720195858]}
    for kk, w in want.items():
        assert np.array_equal(X[:, kk], np.array(w))
    z = O.cartesian_euler_step(np.array([1.0, 2.8, 0.1, 5.0]), 0.3, 0.05, O.Params())
    assert np.array_equal(z, np.array([1.496097858923633, 2.8623451230762056, 0.10559575521166592, 5.03]))


def test_value_net_forward_matches_reference(golden_dir):
    v = np.load(f'{golden_dir}/value_net_golden.npz')
    for sc in (1, 3):
        layers, i = [], 0
        while f'sc{sc}_W{i}' in v:
            layers.append((v[f'sc{sc}_W{i}'], v[f'sc{sc}_b{i}']))
            i += 1
        assert len(layers) == (3 if sc == 1 else 4)
        out = O.value_net_forward(layers, v['z'])[:, 0]
        assert np.abs(out - v[f'V_sc{sc}']).max() < 1e-13


def test_c_restatement_matches_numpy():
    import c_oracle as CO
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.scenarios import make_batch
    A, b = cinf_halfplanes()
    bt = make_batch(48, dtype=np.float64)
    P = O.Params()
    a = (bt['x0'], bt['u_prev'], bt['kparams'], bt['flags'], bt['obs_xy'], A, b, P)
    ref = O.solve_batch(*a, return_all=True)
    allc = CO.rollout_all(*a, nthreads=4)
    assert np.array_equal(allc['U'], ref['U'])
    assert np.abs(allc['X'] - ref['X']).max() < 1e-12
    assert np.array_equal(allc['viol'], ref['mask'].astype(np.uint32))
    got = CO.solve_batch(*a, nthreads=4)
    assert np.array_equal(got['argmin'], ref['argmin']) and np.array_equal(got['status'], ref['status'])
    ok = ref['status'] == 0
    assert ok.any() and (~ok).any()
    assert np.abs(got['x'][ok] - ref['x'][ok]).max() < 1e-12
    assert np.isnan(got['x'][~ok]).all() and np.isinf(got['cost'][~ok]).all()


def test_c_restatement_on_golden(golden_dir):
    import c_oracle as CO
    g = np.load(f'{golden_dir}/frenet_rk4_golden.npz')
    P = O.Params()
    sl = slice(0, 64)
    out = CO.rollout_all(g['x0'][sl], np.zeros((64, 2)), g['kp'][sl], np.zeros(64, np.uint32), None, None, None, P,
                         C=64, U=g['U'][sl])
    got = out['X'][np.arange(64), np.arange(64)]
    assert np.abs(got - g['X'][sl]).max() < 1e-12


def test_oracle_solve_edge_cases():
    from igtmpc.scenarios import make_batch
    P = O.Params()
    bt = make_batch(8, dtype=np.float64)
    bt['x0'][:, 3] = 0.5                                   # every candidate violates |ey| <= 0.2 at k = 0
    r = O.solve_batch(bt['x0'], bt['u_prev'], bt['kparams'], bt['flags'], bt['obs_xy'], None, None, P)
    assert (r['status'] == 1).all() and (r['argmin'] == -1).all() and np.isnan(r['x']).all()
    # tie-break: with identical candidates (table of copies) the lowest index wins
    bt = make_batch(4, dtype=np.float64)
    U = np.zeros((64, 2, 20))
    r = O.solve_batch(bt['x0'], np.zeros((4, 2)), bt['kparams'], bt['flags'], None, None, None, P, C=64, U=U,
                      return_all=True)
    assert ((r['argmin'] == 0) | (r['status'] == 1)).all()
    # abs-heading flag (mpc.py:231-234)
    x = O.apply_flags(np.array([[0, 0, 0, 0, 0, 1, -3.0]]), np.array([1], dtype=np.uint32))
    assert x[0, 6] == 3.0


def test_nlp_quality_yardstick_runs():
    """oracle/nlp_quality.py (scipy SLSQP on the single-shooting restatement of the mpc.py NLP): polishing the
    shooting winner never raises the cost and stays feasible.  Quality metric, not parity."""
    import nlp_quality as Q
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.scenarios import make_batch
    P = O.Params()
    cinf = cinf_halfplanes()
    b = make_batch(16, dtype=np.float64)
    sol = O.solve_batch_refined(b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'], *cinf, P, refine_iters=1)[-1]
    idx = [i for i in range(16) if sol['status'][i] == 0][:2]
    rows = Q.gap_report(b, sol, cinf, P, idx)
    assert len(rows) >= 1
    assert (rows[:, 3] > -1e-6).all() and (rows[:, 3] < 1.5).all()
    assert np.allclose(rows[:, 1], sol['cost'][rows[:, 0].astype(int)], atol=1e-9)   # same cost function


def test_oracle_closed_loop_and_warm_start_properties():
    """oracle/closed_loop.py (restatement of evaluate.py:451-569): runs, exercises fallback / stop / sharing / warm
    start, and the warm-started ramp-hold family contains the shifted previous plan."""
    import closed_loop as CL
    P = O.Params(N=10)
    cinf = (None, None)
    # straight crossing pair; agent 0 starts outside the lane bound and slow -> brake fallback, then the v < 0 stop
    x = np.array([[3.0, 2.8 + 0.25, 3.0, 0.25, 0.0, 0.3, 0.0],
                  [27.8, 40.0, 7.7 + 13.0, 0.0, 0.0, 3.0, -np.pi / 2]])
    r = CL.run_episode(x, ('13', '24'), P, cinf, M_sim=6, cand_mode='ramp_hold', C=64)
    ev = r['events']
    assert ev['fallback'] >= 2 and ev['stop'] >= 1 and ev['share'] >= 4 and ev['warm'] >= 4
    assert r['infeasible'][0] == 6 and r['infeasible'][1] == 0
    assert r['x_data'][5, 1] < 0 and r['x_data'][5, 3] == 0.0          # braked below zero, then frozen at v = 0
    assert np.all(np.diff(r['x_data'][7 + 2]) > 0)                      # the other agent keeps going
    # the shifted previous plan is candidate (G/2, G/2) of the warm-started family
    u_prev = np.array([[0.2, 0.01]])
    prev = O.candidates_lattice(u_prev, P, 64)[0, 37]
    ws = O.shift_controls(prev)
    C, S = O.ramp_hold_first_params(prev[None, :, 0], P)
    U = O.candidates_ramp_hold(prev[None, :, 0], C, S, True, P, 64, ws[None], np.array([True]))
    assert np.abs(U[0, 4 * 8 + 4] - ws).max() < 1e-15
    xa, ua = CL.augment_prev_sol(np.zeros((7, 11)) + np.array([0, 0, 0, 0, 0, 4.99, 0])[:, None], np.full((2, 10), 0.5),
                                 np.array([np.inf, np.inf, 0.0]), P)
    assert xa.shape == (7, 11) and ua.shape == (2, 10) and xa[5, -1] <= 5.0 and abs(xa[5, -1] - 4.99) < 1e-12   # a = 0 retry


def test_tracking_envelope_is_the_stationary_acceleration_of_the_longitudinal_problem():
    """np_oracle.track_env_slope: E_k = dt^2 (N - k - 1/2) / (2 w_u) is where d/da_k [w_u a_k^2 - progress] = 0 on a
    straight lane (mpc.py:362, 372) -- checked by finite differences on the oracle's own cost; the tracking candidate
    with the largest offset follows that line wherever the jerk limit lets it; env = 0 gives the constant targets."""
    P = O.Params()
    slope = O.track_env_slope(P)
    assert slope == P.dt * P.dt / (2 * P.w_u) and O.track_env_slope(P, 0.0) == np.inf
    E = slope * (P.N - np.arange(P.N) - 0.5)
    x0 = np.array([0.0, 0.0, 5.0, 0.0, 0.0, 1.0, 0.0])
    kp = np.array([np.inf, np.inf, 0.0])

    def J(a):
        U = np.stack([a, np.zeros(P.N)])
        return float(O.stage_cost(O.rollout_frenet(x0, U, kp, P), U, P))

    for k in (0, 7, 19):
        g = [(J(E + h * np.eye(P.N)[k]) - J(E - h * np.eye(P.N)[k])) / (2 * h) for h in (1e-4,)][0]
        assert abs(g) < 1e-8, (k, g)
    u_prev = np.zeros((1, 2))
    c, sp = O.track_first_params(u_prev, P)
    U = O.candidates_track(x0[None], u_prev, kp[None], c, sp, True, P, env_slope=slope)[0]
    a = U[15 * 16 + 8, 0]                                   # largest acceleration offset, zero steering offset
    ramp = P.dt * P.jerk * (np.arange(P.N) + 1)
    k_meet = int(np.argmax(ramp > E))                       # first step at which the ramp would overshoot the line
    assert np.allclose(a[:k_meet], ramp[:k_meet], atol=1e-15) and k_meet > 5
    # past the meeting point it comes down at the jerk limit (0.09 per step) towards the line (0.1 per step)
    assert np.all(np.diff(a[k_meet:]) < 0) and np.all(a[k_meet + 1:] - E[k_meet + 1:] < 0.12) and np.all(a[k_meet:] >= E[k_meet:] - 1e-12)
    U0 = O.candidates_track(x0[None], u_prev, kp[None], c, sp, True, P)[0]
    assert np.allclose(U0[15 * 16 + 8, 0], np.minimum(ramp, O.cand_m(15, 16, True) * P.N * P.dt * P.jerk))


def test_tracking_speed_cap_arrives_at_v_max_with_zero_acceleration():
    """np_oracle.track_speed_cap: the largest a_k from which a jerk-limited ramp down (mpc.py:301-304) keeps v <= v_max
    (mpc.py:316-317).  (1) the closed form inverts S(a) = (n + 1) a - r n (n + 1) / 2, n = floor(a / r); (2) rolled forward at
    the cap the speed climbs to v_max and never passes it, the acceleration comes down by at most r per step and ends at 0;
    (3) the tracking candidate with the largest offset follows it -- it survives the speed box where, without the cap, it
    fails it -- and with vcap off the family is the one of before."""
    P = O.Params()
    r = P.dt * P.jerk
    v = np.linspace(P.v_max - 3.0, P.v_max + 0.05, 4001)
    a = O.track_speed_cap(v, P)
    vm = P.v_max - O.TRACK_VCAP_MARGIN
    D = (vm - v) / P.dt
    n = np.floor(np.maximum(a, 0.0) / r + 1e-12)
    S = np.where(D < 0, a, (n + 1) * a - r * n * (n + 1) / 2)
    assert np.abs(S - D).max() < 1e-12 and np.all(np.diff(a) < 0)
    assert np.isinf(O.track_speed_cap(v, P, on=False)).all()
    for v0, a0 in ((3.0, 0.0), (4.2, 0.9), (4.9, 0.3)):
        vv, aa, hist = v0, a0, []
        for k in range(60):
            aa = float(np.clip(aa + np.clip(min(P.a_max, O.track_speed_cap(vv, P)) - aa, -r, r), P.a_min, P.a_max))
            vv = vv + P.dt * aa
            hist.append((vv, aa))
        hv, ha = np.array(hist).T
        assert hv.max() <= vm + 1e-12 and abs(hv[-1] - vm) < 1e-9 and abs(ha[-1]) < 1e-9, (v0, a0, hv.max(), ha[-1])
        assert np.all(np.abs(np.diff(ha)) <= r + 1e-12)
    x0 = np.array([[0.0, 0.0, 5.0, 0.0, 0.0, 4.2, 0.0]])
    kp = np.array([[np.inf, np.inf, 0.0]])
    u_prev = np.array([[0.5, 0.0]])
    c, sp = O.track_first_params(u_prev, P)
    U1 = O.candidates_track(x0, u_prev, kp, c, sp, True, P)[0]
    U0 = O.candidates_track(x0, u_prev, kp, c, sp, True, P, vcap=False)[0]
    top = 15 * 16 + 8
    v1 = x0[0, 5] + np.concatenate([[0.0], np.cumsum(P.dt * U1[top, 0])])
    v0_ = x0[0, 5] + np.concatenate([[0.0], np.cumsum(P.dt * U0[top, 0])])
    assert v1.max() <= vm + 1e-12 and v1[-1] > vm - 1e-6 and v0_.max() > P.v_max + 0.1
    assert abs(U1[top, 0, -1]) < 1e-6
    low = 0 * 16 + 8                                       # a braking row never meets the cap: unchanged
    assert np.array_equal(U1[low], U0[low])


def test_all_eight_shipped_value_networks_reproduce_the_reference_forward(golden_dir):
    """igtmpc/data/value_nets.npz (weights of V_GT_sc1..8, exported as data) through the oracle's forward pass gives the
    outputs the reference's own model.py produced on the golden inputs (tests/golden/make_golden.py, V_sc{n}); the two
    networks that also travel inside the golden file are the same arrays."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'igt-mpc-int_amd'))
    from igtmpc.value_nets import shipped_value_net
    g = np.load(f'{golden_dir}/value_net_golden.npz')
    hidden = {}
    for sc in range(1, 9):
        net = shipped_value_net(sc)
        hidden[sc] = len(net['layers']) - 1
        got = O.value_net_forward(net['layers'], g['z'])
        assert np.abs(got.reshape(-1) - g[f'V_sc{sc}']).max() <= 1e-12, sc
    assert hidden == {1: 2, 2: 2, 3: 3, 4: 2, 5: 2, 6: 3, 7: 3, 8: 2}
    for sc in (1, 3):
        for i, (W, b) in enumerate(shipped_value_net(sc)['layers']):
            assert np.array_equal(W, g[f'sc{sc}_W{i}']) and np.array_equal(b, g[f'sc{sc}_b{i}'])
    with pytest.raises(ValueError):
        shipped_value_net(9)
