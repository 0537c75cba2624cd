"""-m gpu : the reference-shaped Python surface (MPC_Planner, model objects, predictor, closed loop)
runs on the HIP path and agrees with the oracle."""
import os
import sys

import numpy as np
import pytest

import np_oracle as O
from helpers import REL_TOL, rel_err

pytestmark = pytest.mark.gpu


def test_frenet_step_entry_on_golden(golden_dir):
    """igt_frenet_step_* against the reference-generated one-step transitions."""
    import igtmpc
    g = np.load(f'{golden_dir}/frenet_rk4_golden.npz')
    x = g['X'][:, :, 3]
    u = g['U'][:, :, 3]
    want = g['X'][:, :, 4]
    with igtmpc.BatchSolver(dtype='f64', N=1, C=64, n_obs=0) as s:
        got = s.frenet_step(x, u, g['kp'])
    assert rel_err(got, want).max() < 1e-12
    with igtmpc.BatchSolver(dtype='f32', N=1, C=64, n_obs=0) as s:
        got = s.frenet_step(x.astype(np.float32), u.astype(np.float32), g['kp'].astype(np.float32))
    ref = O.frenet_rk4_step(x.astype(np.float32).astype(np.float64), u[:, 0].astype(np.float32).astype(np.float64),
                            u[:, 1].astype(np.float32).astype(np.float64), g['kp'].astype(np.float32).astype(np.float64),
                            O.Params())
    bp = np.minimum(np.abs(ref[:, 2] - g['kp'][:, 0]), np.abs(ref[:, 2] - g['kp'][:, 1]))
    assert rel_err(got, ref)[bp > 1e-2].max() < 2e-6


def test_model_objects_match_reference_models(golden_dir):
    import igtmpc
    g = np.load(f'{golden_dir}/frenet_rk4_golden.npz')
    m = igtmpc.KinematicBicycleModelFrenet(2.235, 2.235, 2.0, 0.1, discretization='rk4', mode='numpy', num_rk4_steps=4)
    for i in (0, 1, 5, 17):
        K = igtmpc.Curvature(*g['kp'][i])
        st = igtmpc.VehicleReference(dict(zip(('x', 'y', 's', 'ey', 'epsi', 'v', 'heading'), g['x0'][i]), K=K))
        for k in range(3):
            st = m(st, igtmpc.VehicleAction({'a': g['U'][i, 0, k], 'df': g['U'][i, 1, k]}))
        assert rel_err(st.state7(), g['X'][i, :, 3]).max() < 1e-12
        assert st.K is K
    c = np.load(f'{golden_dir}/cartesian_euler_golden.npz')
    mc = igtmpc.KinematicBicycleModel(2.235, 2.235, 2.0, 0.1)
    st = igtmpc.VehicleState({'x': c['z'][0, 0], 'y': c['z'][0, 1], 'heading': c['z'][0, 2], 'v': c['z'][0, 3]})
    nx = mc(st, igtmpc.VehicleAction({'a': c['u'][0, 0], 'df': c['u'][0, 1]}))
    assert rel_err([nx.x, nx.y, nx.heading, nx.v], c['z_next'][0]).max() < 1e-12
    # any callable K(s) of the reference's shape is accepted (evaluate.py:384-402 hands the model a casadi Function):
    # it is probed once for its break-points, which come out as the exact float64 numbers it compares against
    i = 1
    b0, b1, kv = (float(q) for q in g['kp'][i])
    plain = lambda s: (kv if s >= b0 else 0.0) - (kv if s >= b1 else 0.0)
    st = {K: igtmpc.VehicleReference(dict(zip(('x', 'y', 's', 'ey', 'epsi', 'v', 'heading'), g['x0'][i]), K=K))
          for K in (plain, igtmpc.Curvature(b0, b1, kv))}
    for k in range(3):
        st = {K: m(v, igtmpc.VehicleAction({'a': g['U'][i, 0, k], 'df': g['U'][i, 1, k]})) for K, v in st.items()}
    a, b = (v.state7() for v in st.values())
    assert a == b and rel_err(a, g['X'][i, :, 3]).max() < 1e-12
    assert igtmpc.Curvature.from_callable(plain).kparams == (b0, b1, kv)
    zero = m(igtmpc.VehicleReference(dict(x=0, y=0, s=0, ey=0, epsi=0, v=1, heading=0, K=lambda s: 0)), igtmpc.VehicleAction({'a': 0, 'df': 0}))
    assert abs(zero.s - 0.1) < 1e-12
    with pytest.raises(TypeError):
        m(igtmpc.VehicleReference(dict(x=0, y=0, s=0, ey=0, epsi=0, v=1, heading=0, K='left')), igtmpc.VehicleAction({'a': 0, 'df': 0}))


def _scene():
    import igtmpc
    from igtmpc import routes as R
    routes = ['13', '23']
    agents, refs = [], []
    for r, s0, v0 in zip(routes, (12.0, 15.0), (3.0, 2.5)):
        rid = R.ROUTE_ID[r]
        xy = R.frenet2global(rid, s0)
        K = igtmpc.Curvature.from_route(r)
        agents.append({'type': 'CAV', 'state': igtmpc.VehicleReference(
            {'x': xy[0], 'y': xy[1], 's': s0, 'ey': 0.02, 'epsi': -0.01, 'v': v0, 'heading': float(R.psi_ref(rid, s0)), 'K': K})})
        kk = np.zeros(151)
        if R.CONSTANTS[r].get('Kv'):
            kk[40:80] = R.CONSTANTS[r]['Kv']
        refs.append({'K': kk})
    return routes, agents, refs


def test_mpc_planner_call_sequence_matches_oracle():
    """update_initial_condition -> update_predictions -> solve, as evaluate.py:470-482 drives it."""
    import igtmpc
    from igtmpc import routes as R
    routes, agents, refs = _scene()
    N = 20
    pred = igtmpc.ConstantAccelerationModel(N=N, dt=0.1)
    inputs = [igtmpc.VehicleAction({'a': 0.1, 'df': 0.0}) for _ in routes]
    preds = pred.predict(agents, inputs, routes, refs)
    assert len(preds) == 2 and len(preds[0]) == N + 1 and preds[0][0].x == agents[0]['state'].x
    P = O.Params(N=N)
    for i in range(2):
        pl = igtmpc.MPC_Planner(N=N, dt=0.1, ca_radius=2.8, agents=agents, routes=routes, ref=refs, goals=None,
                                road_dim=(11.4, 50), ds_right=8.6, index=i, num_rk4_steps=4, dtype='f64',
                                cand_mode='lattice')
        pl.update_initial_condition(agents[i], inputs[i])
        pl.update_predictions(preds, raw_preds=preds)
        x, u, ok = pl.solve()
        assert pl.solve_time > 0 and pl.NN_query_time == -1 and 't_wall_total' in pl.sol.stats()
        if ok:
            assert abs(pl.cost_function() - pl.CAV_utility(x, u)) < 1e-9
        st = agents[i]['state']
        x0 = np.array([st.state7()])
        obs = np.array([[[[p.x for p in preds[1 - i]], [p.y for p in preds[1 - i]]]]])
        ref = O.solve_batch(x0, np.array([[0.1, 0.0]]), np.array([pl.K.kparams]), np.array([0], np.uint32), obs,
                            *pl.C_inf, P)
        assert ok == (ref['status'][0] == 0)
        if ok:
            assert x.shape == (7, N + 1) and u.shape == (2, N)
            assert rel_err(x, ref['x'][0]).max() < 1e-9 and rel_err(u, ref['u'][0]).max() < 1e-12
        else:
            assert x is None and u is None
        # kparams derived from the reference-path curvature array agree with the route table
        assert np.allclose(pl.K.kparams, R.kparams(R.ROUTE_ID[routes[i]]), rtol=1e-12, equal_nan=True)


def test_compat_module_names_import():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, 'igt-mpc-int_amd', 'igtmpc', 'compat'))
    try:
        from mpc import MPC_Planner                                   # noqa: F401
        from VehicleReference import VehicleReference                 # noqa: F401
        from kinematic_bicycle_model_frenet import KinematicBicycleModelFrenet  # noqa: F401
        from constant_acceleration_model import ConstantAccelerationModel      # noqa: F401
        from PredictorBase import PredictorBase                       # noqa: F401
    finally:
        sys.path.pop(0)


def test_closed_loop_short_run():
    from igtmpc.evaluate import run_closed_loop
    r = run_closed_loop(sc=3, num_samples=6, N=20, T_sim=3.0)
    assert r['x_data'].shape == (6, 14, 31) and np.isfinite(r['x_data']).all()
    s = r['x_data'][:, 2::7, :]
    assert (np.diff(s, axis=-1) >= -1e-6).all(), 'vehicles never reverse along their route'
    assert (s[:, :, -1] > s[:, :, 0] + 1.0).all(), 'straight-crossing vehicles make progress in 3 s'
    v = r['x_data'][:, 5::7, :]
    assert v.max() <= 5 + 1e-3 and v.min() >= -1e-3
    assert np.abs(r['x_data'][:, 3::7, :]).max() <= 0.2 + 1e-3            # |ey| <= 0.2 held in closed loop
    # same seed -> same episodes
    r2 = run_closed_loop(sc=3, num_samples=6, N=20, T_sim=3.0)
    assert np.array_equal(r['x_data'], r2['x_data'])


def test_forecast_matches_oracle(golden_dir):
    """igt_forecast_batch_* (constant-acceleration forecast, V2V plan sharing, filter_preds) against the
    route-by-route oracle restatement; both branches (raw forecast / shared plan) and both filter outcomes."""
    import json
    import igtmpc
    from igtmpc import routes as R
    with open(f'{golden_dir}/route_constants.json') as f:
        C = json.load(f)
    rng = np.random.default_rng(11)
    B, N, dt = 192, 20, 0.1
    rid = rng.integers(0, 12, B)
    s0 = rng.uniform(0, 50, B)
    v0 = rng.uniform(0, 5.5, B)
    a0 = rng.uniform(-2, 2, B)
    xy = R.frenet2global(rid, s0)
    opp = np.column_stack([xy, s0, v0])
    ego = np.column_stack([rng.uniform(0, 50, B), rng.uniform(-20, 30, B), rng.uniform(-np.pi, np.pi, B)])
    has_plan = (rng.random(B) < 0.5).astype(np.int32)
    plan_x = rng.uniform(-5, 45, (B, 7, N + 1))
    plan_x[:, 5, :] = rng.uniform(3.5, 5.2, (B, N + 1))              # v near the 5 m/s retry threshold
    plan_u = rng.uniform(-1, 2, (B, 2, N))
    for dtype, npdt, tol in (('f64', np.float64, 1e-12), ('f32', np.float32, REL_TOL)):
        with igtmpc.BatchSolver(dtype=dtype, N=N) as s:
            obs, tv = s.forecast(ego.astype(npdt), opp.astype(npdt), a0.astype(npdt), rid.astype(np.int32),
                                 plan_x.astype(npdt), plan_u.astype(npdt), has_plan)
            obs_raw, _ = s.forecast(ego.astype(npdt), opp.astype(npdt), a0.astype(npdt), rid.astype(np.int32))
        n_filtered = 0
        for b in range(B):
            cast = lambda z: np.asarray(z, dtype=npdt).astype(np.float64)
            st = [cast(opp[b, 0]), cast(opp[b, 1]), cast(opp[b, 2]), 0, 0, cast(opp[b, 3]), 0]
            e = cast(ego[b])
            ref, (sl, vl) = O.forecast_for_ego(R.ROUTES[rid[b]], C[R.ROUTES[rid[b]]], e[:2], e[2], st, float(cast(a0[b])), N, dt,
                                               cast(plan_x[b]) if has_plan[b] else None, cast(plan_u[b]) if has_plan[b] else None)
            assert rel_err(obs[b, 0], ref).max() < tol
            assert rel_err(tv[b], [sl, vl]).max() < tol
            ref_raw, _ = O.forecast_for_ego(R.ROUTES[rid[b]], C[R.ROUTES[rid[b]]], e[:2], e[2], st, float(cast(a0[b])), N, dt)
            assert rel_err(obs_raw[b, 0], ref_raw).max() < tol
            n_filtered += ref[0, 0] == -20.0
        assert 0.2 * B < n_filtered < 0.8 * B


def test_forecast_of_three_and_four_vehicle_scenes_matches_oracle(golden_dir):
    """Scenes of M > 2 vehicles (mpc.py:82-83 is written for any M): igt_forecast_batch_* with n_obs = 2 and 3 -- every
    opponent of a scene forecast / shared / filtered on its own -- equals the two-vehicle entry called once per opponent, and the
    route-by-route oracle; the result is what igt_solve_batch_* takes as obs_xy."""
    import json
    import igtmpc
    from igtmpc import routes as R
    with open(f'{golden_dir}/route_constants.json') as f:
        C = json.load(f)
    rng = np.random.default_rng(12)
    B, N, dt = 96, 20, 0.1
    ego = np.column_stack([rng.uniform(0, 50, B), rng.uniform(-20, 30, B), rng.uniform(-np.pi, np.pi, B)])
    for M1 in (2, 3):
        rid = rng.integers(0, 12, (B, M1))
        s0, v0, a0 = rng.uniform(0, 50, (B, M1)), rng.uniform(0, 5.5, (B, M1)), rng.uniform(-2, 2, (B, M1))
        xy = R.frenet2global(rid.ravel(), s0.ravel()).reshape(B, M1, 2)
        opp = np.concatenate([xy, s0[..., None], v0[..., None]], axis=-1)
        has_plan = (rng.random((B, M1)) < 0.5).astype(np.int32)
        plan_x = rng.uniform(-5, 45, (B, M1, 7, N + 1))
        plan_x[:, :, 5, :] = rng.uniform(3.5, 5.2, (B, M1, N + 1))
        plan_u = rng.uniform(-1, 2, (B, M1, 2, N))
        for dtype, npdt, tol in (('f64', np.float64, 1e-12), ('f32', np.float32, REL_TOL)):
            c = lambda z: np.ascontiguousarray(z, dtype=npdt)
            with igtmpc.BatchSolver(dtype=dtype, N=N, n_obs=M1) as s:
                obs, tv = s.forecast(c(ego), c(opp), c(a0), rid.astype(np.int32), c(plan_x), c(plan_u), has_plan)
            assert obs.shape == (B, M1, 2, N + 1) and tv.shape == (B, M1, 2)
            with igtmpc.BatchSolver(dtype=dtype, N=N, n_obs=1) as s1:
                for o in range(M1):
                    o1, t1 = s1.forecast(c(ego), c(opp[:, o]), c(a0[:, o]), rid[:, o].astype(np.int32), c(plan_x[:, o]), c(plan_u[:, o]),
                                         np.ascontiguousarray(has_plan[:, o]))
                    assert np.array_equal(o1[:, 0], obs[:, o]) and np.array_equal(t1, tv[:, o])
            cast = lambda z: np.asarray(z, dtype=npdt).astype(np.float64)
            for b in range(0, B, 5):
                for o in range(M1):
                    st = [cast(opp[b, o, 0]), cast(opp[b, o, 1]), cast(opp[b, o, 2]), 0, 0, cast(opp[b, o, 3]), 0]
                    e = cast(ego[b])
                    ref, (sl, vl) = O.forecast_for_ego(R.ROUTES[rid[b, o]], C[R.ROUTES[rid[b, o]]], e[:2], e[2], st, float(cast(a0[b, o])), N, dt,
                                                       cast(plan_x[b, o]) if has_plan[b, o] else None, cast(plan_u[b, o]) if has_plan[b, o] else None)
                    assert rel_err(obs[b, o], ref).max() < tol and rel_err(tv[b, o], [sl, vl]).max() < tol


def test_closed_loop_gt_mpc_mode(golden_dir):
    """evaluate.py --eval_mode gt_mpc counterpart: value network in the cost, (0,0) initial inputs, the first
    step's special forecast; runs with the shipped sc1 checkpoint (identity normalisation)."""
    from igtmpc.evaluate import run_closed_loop
    v = np.load(f'{golden_dir}/value_net_golden.npz')
    layers, i = [], 0
    while f'sc1_W{i}' in v:
        layers.append((v[f'sc1_W{i}'], v[f'sc1_b{i}']))
        i += 1
    r = run_closed_loop(sc=1, num_samples=4, N=20, T_sim=2.0, eval_mode='gt_mpc', value_net=dict(layers=layers))
    assert r['x_data'].shape == (4, 14, 21) and np.isfinite(r['x_data']).all()
    m = run_closed_loop(sc=1, num_samples=4, N=20, T_sim=2.0)
    assert not np.array_equal(r['u_data'], m['u_data'])          # the terminal value changes the decisions
    # without value_net the driver takes the network the reference ships for the scenario (sc1_config.yaml:2)
    d = run_closed_loop(sc=1, num_samples=4, N=20, T_sim=2.0, eval_mode='gt_mpc')
    assert np.array_equal(d['x_data'], r['x_data']) and np.array_equal(d['u_data'], r['u_data'])
    r6 = run_closed_loop(sc=6, num_samples=2, N=20, T_sim=1.0, eval_mode='gt_mpc')        # a 3-hidden-layer one
    assert np.isfinite(r6['x_data']).all()


def test_closed_loop_device_resident_equals_host_loop():
    """All per-step arrays in HBM (torch) vs the numpy loop: same kernels, same decisions."""
    from igtmpc.evaluate import run_closed_loop
    for sc, T in ((4, 2.5), (1, 4.0)):
        a = run_closed_loop(sc=sc, num_samples=8, N=20, T_sim=T)
        b = run_closed_loop(sc=sc, num_samples=8, N=20, T_sim=T, device_resident=True)
        # (also guards the stream ordering of device-mode calls against the surrounding torch ops)
        assert np.array_equal(a['x_data'], b['x_data']) and np.array_equal(a['u_data'], b['u_data'])
        assert np.array_equal(a['infeasible_ratio'], b['infeasible_ratio'])


def test_closed_loop_replayed_from_a_stream_graph_equals_the_eager_loop(golden_dir):
    """device_resident + graph: after two eager steps ONE step (forecast, solve, fallback step and the tensor operations
    between them) is captured and replayed for the rest of the episode; trajectories, applied inputs and infeasible
    counts are those of the eager device loop bit for bit -- tracking candidates + warm start, the lattice, and the
    gt_mpc loop (value network inside the captured solve)."""
    from igtmpc.evaluate import run_closed_loop
    v = np.load(f'{golden_dir}/value_net_golden.npz')
    layers, i = [], 0
    while f'sc1_W{i}' in v:
        layers.append((v[f'sc1_W{i}'], v[f'sc1_b{i}']))
        i += 1
    for kw in (dict(sc=2, cand_mode='track', warm_start=True), dict(sc=2, cand_mode='track'), dict(sc=7, cand_mode='lattice'),
               dict(sc=1, cand_mode='track', warm_start=True, eval_mode='gt_mpc', value_net=dict(layers=layers))):
        a = run_closed_loop(num_samples=8, N=20, T_sim=3.0, device_resident=True, **kw)
        b = run_closed_loop(num_samples=8, N=20, T_sim=3.0, device_resident=True, graph=True, **kw)
        assert np.array_equal(a['x_data'], b['x_data']) and np.array_equal(a['u_data'], b['u_data']), kw
        assert np.array_equal(a['infeasible_ratio'], b['infeasible_ratio']) and np.array_equal(a['deadlock'], b['deadlock'])
        assert np.abs(b['u_data'][:, :, -1]).max() > 0          # the last column was written by the last replay


def test_mpc_planner_gt_mode(golden_dir):
    """use_NN_cost2go=True (evaluate.py:191): the planner drives the value-net cost through the same call
    sequence and agrees with the oracle."""
    import igtmpc
    from igtmpc import routes as R
    v = np.load(f'{golden_dir}/value_net_golden.npz')
    layers, i = [], 0
    while f'sc1_W{i}' in v:
        layers.append((v[f'sc1_W{i}'], v[f'sc1_b{i}']))
        i += 1
    net = dict(layers=layers, Wn=np.eye(6), mu_f=np.zeros(6), sigma_t=1.0, mu_t=0.0)
    routes, agents, refs = _scene()
    N = 20
    pred = igtmpc.ConstantAccelerationModel(N=N, dt=0.1)
    inputs = [igtmpc.VehicleAction({'a': 0.0, 'df': 0.0}) for _ in routes]
    preds = pred.predict(agents, inputs, routes, refs)
    P = O.Params(N=N)
    with pytest.raises(ValueError):
        igtmpc.MPC_Planner(N=N, dt=0.1, agents=agents, routes=routes, ref=refs, road_dim=(11.4, 50), ds_right=8.6,
                           index=0, num_rk4_steps=4, use_NN_cost2go=True)            # no statistics / weights given
    for i in range(2):
        pl = igtmpc.MPC_Planner(N=N, dt=0.1, ca_radius=2.8, agents=agents, routes=routes, ref=refs, goals=None,
                                road_dim=(11.4, 50), ds_right=8.6, index=i, num_rk4_steps=4, dtype='f64',
                                use_NN_cost2go=True, value_net=dict(net), cand_mode='lattice')
        pl.update_initial_condition(agents[i], inputs[i])
        pl.update_predictions(preds, raw_preds=preds)
        x, u, ok = pl.solve()
        st = agents[i]['state']
        j = 1 - i
        obs = np.array([[[[p.x for p in preds[j]], [p.y for p in preds[j]]]]])
        e = R.scenario_encoding_sign(routes, R.scenario_of(routes))
        ref = O.solve_batch(np.array([st.state7()]), np.zeros((1, 2)), np.array([pl.K.kparams]), np.array([0], np.uint32),
                            obs, *pl.C_inf, P, net=net, tv_sv=np.array([[preds[j][-1].s, preds[j][-1].v]]),
                            enc=np.array([[e[i], e[j]]], dtype=np.float64))
        assert ok == (ref['status'][0] == 0)
        if ok:
            assert rel_err(x, ref['x'][0]).max() < 1e-9


def test_nan_inputs_and_determinism():
    import igtmpc
    from igtmpc.scenarios import make_batch
    b = make_batch(64, dtype=np.float32)
    bad = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in b.items()}
    bad['x0'][3, 2] = np.nan          # NaN arc length
    bad['x0'][5, 5] = np.inf          # infinite speed
    with igtmpc.BatchSolver(dtype='f32') as s:
        o1 = s.solve(b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])
        o2 = s.solve(b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])
        ob = s.solve(bad['x0'], bad['u_prev'], bad['kparams'], bad['flags'], bad['obs_xy'])
    for k in o1:
        assert np.array_equal(o1[k], o2[k], equal_nan=True)       # run-to-run bitwise reproducible
    assert ob['status'][3] == 1 and ob['status'][5] == 1            # non-finite problems are reported, not propagated
    keep = np.ones(64, bool)
    keep[[3, 5]] = False
    for k in o1:
        assert np.array_equal(ob[k][keep], o1[k][keep], equal_nan=True)

# ----------------------------------------------------------------------------- stream capture
@pytest.mark.parametrize('B,dtype,cand', [(512, 'f32', 'lattice'), (4096, 'f32', 'lattice'), (4096, 'f64', 'lattice'),
                                          (2048, 'f64', 'track')])
def test_solve_is_capturable_in_a_graph(B, dtype, cand):
    """The solve is a fixed sequence of stream operations (kernels, one memset node) once its workspace exists: captured
    in a graph and replayed on new inputs written into the same buffers, it gives what the eager call gives.
    (B = 512: queue builder + checkpointed emit; B = 4096: queue builder + plain emit; f64: acceleration rows, queue builder,
    search with the incumbents of the tracking family, emit in pieces.  The graph is replayed three times, on alternating
    inputs, with eager solves of another size on the same handle in between: every launch must leave the handle's counters
    and incumbents as it found them.)"""
    import torch
    import igtmpc
    from igtmpc.scenarios import make_batch
    from igtmpc.cinf import cinf_halfplanes
    npdt = np.float32 if dtype == 'f32' else np.float64

    def dev(b, n=None):
        return [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a)[:n].contiguous().cuda()
                for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]

    b1, b2 = make_batch(B, dtype=npdt, seed=1), make_batch(B, dtype=npdt, seed=2)
    with igtmpc.BatchSolver(dtype=dtype, cand_mode=cand) as s:
        s.set_cinf(*cinf_halfplanes())
        bufs = dev(b1)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            out = s.solve(*bufs)                     # warm-up on the capture stream: the workspace is allocated here
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            s.solve(*bufs, out=out)
        for rnd, src_batch in enumerate((b2, b1, b2)):
            for dst, src in zip(bufs, dev(src_batch)):
                dst.copy_(src)
            g.replay()
            torch.cuda.synchronize()
            replayed = {k: v.clone() for k, v in out.items()}
            smaller = s.solve(*dev(src_batch, B - 300))      # (a smaller batch: the workspace does not grow)
            eager = s.solve(*dev(src_batch))
            torch.cuda.synchronize()
            for k in ('x', 'u', 'cost', 'argmin', 'status'):
                assert torch.equal(replayed[k].nan_to_num(), eager[k].nan_to_num()), (rnd, k)
                assert torch.equal(smaller[k].nan_to_num(), eager[k][:B - 300].nan_to_num()), (rnd, k)
    assert (eager['status'] == 0).float().mean() > 0.5


@pytest.mark.parametrize('how', ['default_stream_object', 'zero', 'implicit'])
def test_explicit_default_stream_is_ordered_with_torch_ops(how):
    """An explicitly passed default stream (handle 0) must mean torch's legacy default stream, not the handle's own
    non-blocking stream: the solve is enqueued right after an in-place update of its inputs on that stream and read
    right after it, without any synchronisation in between; it must see the update and be seen by the read."""
    import torch
    import igtmpc
    from igtmpc.scenarios import make_batch
    from igtmpc.cinf import cinf_halfplanes
    B = 4096
    b1, b2 = make_batch(B, dtype=np.float32, seed=3), make_batch(B, dtype=np.float32, seed=4)
    keys = ('x0', 'u_prev', 'kparams', 'flags', 'obs_xy')
    T = lambda a: torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a)
    with igtmpc.BatchSolver(dtype='f32') as s:
        s.set_cinf(*cinf_halfplanes())
        want = s.solve(*[b2[k] for k in keys])                      # host mode: synchronous reference result
        bufs = [T(b1[k]).cuda() for k in keys]
        out = s.solve(*bufs)                                        # workspace allocation + warm-up
        torch.cuda.synchronize()
        staged = [T(b2[k]).pin_memory() for k in keys]
        stream = {'default_stream_object': torch.cuda.default_stream(), 'zero': 0, 'implicit': None}[how]
        # a long-running producer on the default stream, then the in-place input update, then the solve
        junk = torch.randn(4096, 4096, device='cuda')
        for _ in range(8):
            junk = junk @ junk * 1e-3
        for dst, src in zip(bufs, staged):
            dst.copy_(src, non_blocking=True)
        s.solve(*bufs, out=out, stream=stream)
        got = {k: v.clone() for k, v in out.items()}                # consumer on the default stream, no sync before
        torch.cuda.synchronize()
    for k in ('x', 'u', 'cost', 'argmin', 'status'):
        assert np.array_equal(got[k].cpu().numpy(), want[k], equal_nan=True), (how, k)


def test_value_net_solve_is_capturable_and_growth_under_capture_fails_loudly(golden_dir):
    """gt_mpc solve (search + MFMA value kernel + atomicMin winner + emit) captured in a graph == eager; and a solve
    that would have to GROW the workspace while the stream is capturing returns IGT_E_STATE instead of synchronising
    a capturing stream (which would invalidate the capture with an opaque HIP error)."""
    import torch
    import igtmpc
    from igtmpc.scenarios import make_batch
    from igtmpc.cinf import cinf_halfplanes
    v = np.load(f'{golden_dir}/value_net_golden.npz')
    layers, i = [], 0
    while f'sc3_W{i}' in v:
        layers.append((v[f'sc3_W{i}'], v[f'sc3_b{i}']))
        i += 1
    keys = ('x0', 'u_prev', 'kparams', 'flags', 'obs_xy', 'tv_sv', 'enc')

    def dev(b, n=None):
        return [torch.from_numpy(b[k].view(np.int32) if b[k].dtype == np.uint32 else b[k])[:n].contiguous().cuda()
                for k in keys]

    B = 4096
    b1, b2 = make_batch(B, dtype=np.float32, seed=1), make_batch(B, dtype=np.float32, seed=2)
    with igtmpc.BatchSolver(dtype='f32', cost_mode='value_net') as s:
        s.set_cinf(*cinf_halfplanes())
        s.set_value_net(layers)
        side = torch.cuda.Stream()
        small = dev(b1, 256)
        with torch.cuda.stream(side):
            out_small = s.solve(*small)              # workspace sized for B = 256 only
        side.synchronize()
        bufs = dev(b1)
        out = {k: torch.empty((B,) + tuple(t.shape[1:]), dtype=t.dtype, device='cuda') for k, t in out_small.items()}
        g0 = torch.cuda.CUDAGraph()
        with pytest.raises(igtmpc.IgtError, match='workspace too small for stream capture'):
            with torch.cuda.graph(g0, stream=side):
                s.solve(*bufs, out=out)
        del g0
        with torch.cuda.stream(side):
            s.solve(*bufs, out=out)                  # eager warm-up at the real size
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            s.solve(*bufs, out=out)
        for dst, src in zip(bufs, dev(b2)):
            dst.copy_(src)
        g.replay()
        torch.cuda.synchronize()
        replayed = {k: t.clone() for k, t in out.items()}
        eager = s.solve(*dev(b2))
        torch.cuda.synchronize()
    for k in ('x', 'u', 'cost', 'argmin', 'status'):
        assert torch.equal(replayed[k].nan_to_num(), eager[k].nan_to_num()), k
    assert (eager['status'] == 0).float().mean() > 0.5


def test_mpc_planner_warm_start_sequence_matches_oracle():
    """The reference's per-step call sequence with the warm start (evaluate.py:470-482): solve, then
    augment_prev_sol -> solve(x_sol_prev, u_sol_prev).  Default planner = ramp-hold candidates; with a warm start they are
    centred on the shifted previous solution, which is then itself candidate (G/2, G/2)."""
    import igtmpc
    import closed_loop as CL
    from igtmpc.planner import augment_prev_sol
    routes, agents, refs = _scene()
    N = 20
    P = O.Params(N=N)
    pred = igtmpc.ConstantAccelerationModel(N=N, dt=0.1)
    inputs = [igtmpc.VehicleAction({'a': 0.1, 'df': 0.0}) for _ in routes]
    preds = pred.predict(agents, inputs, routes, refs)
    model = igtmpc.KinematicBicycleModelFrenet(2.235, 2.235, 2.0, 0.1, discretization='rk4', mode='numpy', num_rk4_steps=4)
    i = 0
    dflt = igtmpc.MPC_Planner(N=N, dt=0.1, ca_radius=2.8, agents=agents, routes=routes, ref=refs, goals=None,
                              road_dim=(11.4, 50), ds_right=8.6, index=i, num_rk4_steps=4)
    assert dflt.cand_mode == 'track' and dflt._solver.dtype == 'f64'      # defaults: tracking candidates, reference precision
    dflt.update_initial_condition(agents[i], inputs[i])
    dflt.update_predictions(preds, raw_preds=preds)
    xd, ud, okd = dflt.solve()
    st0 = agents[i]['state']
    rd = O.solve_batch_refined(np.array([st0.state7()]), np.array([[0.1, 0.0]]), np.array([dflt.K.kparams]), np.array([0], np.uint32),
                               np.array([[[[p.x for p in preds[1]], [p.y for p in preds[1]]]]]), *dflt.C_inf, P, cand='track',
                               track=dict(env=dflt.track_env))[-1]
    assert dflt.track_env == 0.5          # horizon-aware default (igtmpc.evaluate.auto_track_env): N dt = 2 s of 4
    assert okd == (rd['status'][0] == 0) and (not okd or rel_err(xd, rd['x'][0]).max() < 1e-9)
    # the tracking family's default takes solve(x_sol_prev, u_sol_prev) as the reference passes it and does NOT centre its
    # candidates there (warm_start=None: ramp-hold yes, tracking no -- DESIGN section 9); warm_start=True does, as the oracle does
    assert okd and not dflt.warm_start
    hint = np.stack([np.full(N, -1.5), np.zeros(N)])          # (far enough below the envelope that no offset reaches it)
    xh, uh, okh = dflt.solve(x_sol_prev=xd, u_sol_prev=hint)
    assert okh and np.array_equal(xh, xd) and np.array_equal(uh, ud)
    wt = igtmpc.MPC_Planner(N=N, dt=0.1, ca_radius=2.8, agents=agents, routes=routes, ref=refs, goals=None,
                            road_dim=(11.4, 50), ds_right=8.6, index=i, num_rk4_steps=4, warm_start=True)
    wt.update_initial_condition(agents[i], inputs[i])
    wt.update_predictions(preds, raw_preds=preds)
    xw_, uw_, okw = wt.solve(x_sol_prev=xd, u_sol_prev=hint)
    rw = O.solve_batch_refined(np.array([st0.state7()]), np.array([[0.1, 0.0]]), np.array([wt.K.kparams]), np.array([O.FLAG_WARM], np.uint32),
                               np.array([[[[p.x for p in preds[1]], [p.y for p in preds[1]]]]]), *wt.C_inf, P, cand='track',
                               track=dict(env=wt.track_env), u_ws=hint[None])[-1]
    assert wt.warm_start and okw == (rw['status'][0] == 0) and (not okw or rel_err(xw_, rw['x'][0]).max() < 1e-9)
    assert not okw or not np.array_equal(uw_, ud)          # the hint moved the candidates
    pl = igtmpc.MPC_Planner(N=N, dt=0.1, ca_radius=2.8, agents=agents, routes=routes, ref=refs, goals=None,
                            road_dim=(11.4, 50), ds_right=8.6, index=i, num_rk4_steps=4, cand_mode='ramp_hold')
    assert pl.cand_mode == 'ramp_hold' and pl._solver.dtype == 'f64'
    pl.update_initial_condition(agents[i], inputs[i])
    pl.update_predictions(preds, raw_preds=preds)
    x1, u1, ok = pl.solve()
    assert ok
    st = agents[i]['state']
    x0 = np.array([st.state7()])
    obs = np.array([[[[p.x for p in preds[1]], [p.y for p in preds[1]]]]])
    kp = np.array([pl.K.kparams])
    r1 = O.solve_batch_refined(x0, np.array([[0.1, 0.0]]), kp, np.array([0], np.uint32), obs, *pl.C_inf, P)[-1]
    assert rel_err(x1, r1['x'][0]).max() < 1e-9 and rel_err(u1, r1['u'][0]).max() < 1e-12
    # next step: state x1[:,1], applied input u1[:,0], warm start from augment_prev_sol
    xw, uw = augment_prev_sol((x1, u1), model, st.K)
    xo, uo = CL.augment_prev_sol(x1, u1, kp[0], P)
    assert rel_err(xw, xo).max() < 1e-12 and np.array_equal(uw, uo)
    nxt = dict(zip(('x', 'y', 's', 'ey', 'epsi', 'v', 'heading'), x1[:, 1]), K=st.K)
    pl.update_initial_condition({'type': 'CAV', 'state': igtmpc.VehicleReference(nxt)}, igtmpc.VehicleAction({'a': u1[0, 0], 'df': u1[1, 0]}))
    x2, u2, ok2 = pl.solve(x_sol_prev=xw, u_sol_prev=uw)
    r2 = O.solve_batch_refined(x1[None, :, 1], u1[None, :, 0], kp, np.array([O.FLAG_WARM], np.uint32), obs, *pl.C_inf, P,
                               u_ws=uw[None])[-1]
    assert ok2 == (r2['status'][0] == 0)
    if ok2:
        assert rel_err(x2, r2['x'][0]).max() < 1e-9 and rel_err(u2, r2['u'][0]).max() < 1e-12
        # the shifted previous plan is a candidate, so the new cost cannot exceed its cost
        G = 16
        cand = O.candidates_ramp_hold(u1[None, :, 0], np.zeros((1, 2)), np.array([[1.8, 1.4]]), True, P, 256, uw[None], np.array([True]))
        assert rel_err(cand[0, (G // 2) * G + G // 2], uw).max() < 1e-12


@pytest.mark.parametrize('cand_mode', ['lattice', 'ramp_hold', 'track'])
def test_closed_loop_matches_oracle_loop(cand_mode):
    """igtmpc.evaluate (batched, lock-step, GPU, float64 entry points) against oracle/closed_loop.py -- the plain
    per-episode, per-agent restatement of evaluate.py:451-569: forecast -> share (v > 5 retry) -> filter -> warm start
    -> solve -> brake fallback / instantaneous stop -> deadlock.  4 episodes x 30 steps; two sampled like the
    reference samples them, one started fast (hits the shared-plan retry), one started outside the lane bound and slow
    (hits the brake fallback and the v < 0 stop)."""
    import closed_loop as CL
    from igtmpc import routes as R
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.evaluate import initial_states, run_closed_loop
    pairs = [R.SCENARIO_ROUTES[0][0], R.SCENARIO_ROUTES[3][1], ('13', '24'), ('13', '23')]
    x, _ = initial_states(np.random.default_rng(2026), pairs)
    x[0, :, 5] = 2.0                                              # sampled episodes start rolling (v0 = 0 crawls for 3 s)
    x[1, :, 5] = 2.5
    for m, (r, s0, v0) in enumerate(zip(pairs[2], (3.0, 5.0), (4.6, 4.7))):
        xy = R.frenet2global(R.ROUTE_ID[r], s0)
        x[2, m] = (xy[0], xy[1], s0, 0.0, 0.0, v0, float(R.psi_ref(R.ROUTE_ID[r], s0)))
    x[3, 0, 3], x[3, 0, 5] = 0.25, 0.3                            # |ey_0| > 0.2: infeasible from the first step on
    x[3, 1, 5] = 2.0
    P = O.Params(N=20)
    ev = dict(fallback=0, stop=0, share=0, share_retry=0, warm=0)
    # With the terminal set, a feasible plan ends inside C_inf, i.e. at most 0.009 m/s above v = 5 one step later: the
    # shared-plan retry (utils.py:348) is all but unreachable.  The fast episode therefore runs without it -- and, in the tracking
    # family, without the speed cap of the targets (igt_params.track_vcap), under which no plan ends above v = 5 either.
    for sel, terminal in (([0, 1, 3], True), ([2], False)):
        vcap = 1.0 if terminal else 0.0
        got = run_closed_loop(N=20, T_sim=3.0, dtype='f64', cand_mode=cand_mode, init=(x[sel], [pairs[e] for e in sel]),
                              terminal_set=terminal, limits=dict(track_vcap=vcap) if cand_mode == 'track' else None,
                              warm_start=True)      # (the tracking family's default is off: asked for, as the oracle loop has it)
        cinf = cinf_halfplanes() if terminal else (None, None)
        for q, e in enumerate(sel):
            ref = CL.run_episode(x[e], pairs[e], P, cinf, M_sim=30, cand_mode=cand_mode,
                                 track_env=got['track_env'] if cand_mode == 'track' else 1.0, track_vcap=vcap)
            for k in ev:
                ev[k] += ref['events'][k]
            assert rel_err(got['x_data'][q], ref['x_data']).max() < 1e-9, (e, pairs[e])
            assert rel_err(got['u_data'][q], ref['u_data']).max() < 1e-9, (e, pairs[e])
            assert np.array_equal(got['infeasible_ratio'][q] * 30, ref['infeasible']), (e, pairs[e])
            assert bool(got['deadlock'][q]) == ref['deadlock']
    assert ev['fallback'] > 0 and ev['stop'] > 0 and ev['share'] > 0, ev
    # constant-increment lattice plans started at a = 0.1 never end above v = 5 one step past the horizon; ramp-hold
    # plans do (the retry branch itself is also pinned directly in test_forecast_matches_oracle)
    assert (ev['share_retry'] > 0) == (cand_mode != 'lattice'), ev
    assert (ev['warm'] > 0) == (cand_mode != 'lattice')


@pytest.mark.parametrize('cand_mode,N', [('track', 20), ('lattice', 20), ('track', 40)])
def test_closed_loop_gt_mpc_matches_oracle_loop(cand_mode, N):
    """eval_mode='gt_mpc' (evaluate.py:202-330) against oracle/closed_loop.py's gt mode, at 1e-9: (0, 0) initial inputs
    (:171), the first step's forecast with a = 0.09 (k + 1) (:207-210), warm start for t > 1 only (:232), the terminal
    value of (s_tv, v_tv) taken from the shared but UNFILTERED forecast (mpc.py:330), the scenario encodings, and a
    synthetic non-identity whitening / de-normalisation (the reference's statistics are not shipped).  Also the loop at
    the horizon the reference ships (N = 40, mpc.yaml:6)."""
    import closed_loop as CL
    from igtmpc import routes as R, shipped_value_net
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.evaluate import initial_states, run_closed_loop
    rng = np.random.default_rng(5)
    net = dict(shipped_value_net(1), Wn=np.eye(6) + 0.05 * rng.normal(size=(6, 6)),
               mu_f=np.array([20.0, 2.5, 0.0, 0.0, 0.0, 0.0]) + 0.1 * rng.normal(size=6), sigma_t=3.0, mu_t=-1.5)
    pairs = [R.SCENARIO_ROUTES[0][0], R.SCENARIO_ROUTES[0][3], R.SCENARIO_ROUTES[0][1]]      # scenario 1: ('13','23'), ('12','42'), ('24','34')
    x, _ = initial_states(np.random.default_rng(2026), pairs)
    x[0, :, 5], x[1, :, 5] = 2.0, 2.5
    x[2, 0, 3], x[2, 0, 5] = 0.25, 0.3                            # |ey_0| > 0.2: agent 0 falls back from the first step on
    x[2, 1, 5] = 2.0
    M_sim = 24 if N == 20 else 12
    P = O.Params(N=N)
    got = run_closed_loop(N=N, T_sim=M_sim * 0.1, dtype='f64', cand_mode=cand_mode, init=(x, pairs), eval_mode='gt_mpc',
                          value_net=net, warm_start=True)
    dev = run_closed_loop(N=N, T_sim=M_sim * 0.1, dtype='f64', cand_mode=cand_mode, init=(x, pairs), eval_mode='gt_mpc',
                          value_net=net, device_resident=True, warm_start=True)
    assert np.array_equal(got['x_data'], dev['x_data']) and np.array_equal(got['u_data'], dev['u_data'])
    ev = dict(fallback=0, stop=0, share=0, share_retry=0, warm=0)
    for e in range(len(pairs)):
        ref = CL.run_episode(x[e], pairs[e], P, cinf_halfplanes(), M_sim=M_sim, cand_mode=cand_mode, eval_mode='gt_mpc',
                             net=net)
        for k in ev:
            ev[k] += ref['events'][k]
        assert rel_err(got['x_data'][e], ref['x_data']).max() < 1e-9, (e, pairs[e])
        assert rel_err(got['u_data'][e], ref['u_data']).max() < 1e-9, (e, pairs[e])
        assert np.array_equal(got['infeasible_ratio'][e] * M_sim, ref['infeasible']), (e, pairs[e])
        assert bool(got['deadlock'][e]) == ref['deadlock']
    assert ev['fallback'] > 0 and ev['share'] > 0, ev
    assert (ev['warm'] > 0) == (cand_mode == 'track')
    if cand_mode == 'track':      # agents that solved steps 0 and 1 got no warm start at t = 1: at most M_sim - 2 each
        assert ev['warm'] <= 2 * len(pairs) * (M_sim - 2)


def test_gt_mpc_warm_start_begins_at_t2_not_t1():
    """evaluate.py:232, observed from outside: with tracking candidates the first TWO solves of the gt_mpc loop are those
    of a loop run without warm starts, and from t = 2 on its warm start is in use (the rule of both branches is pinned
    step by step in test_closed_loop_*_matches_oracle_loop; tests/test_host_logic.py counts the oracle's warm starts)."""
    from igtmpc.evaluate import run_closed_loop
    kw = dict(sc=1, num_samples=6, N=20, T_sim=0.8, dtype='f64', cand_mode='track')
    for dev in (False, True):
        g_w = run_closed_loop(eval_mode='gt_mpc', warm_start=True, device_resident=dev, **kw)
        g_c = run_closed_loop(eval_mode='gt_mpc', warm_start=False, device_resident=dev, **kw)
        assert np.array_equal(g_w['u_data'][:, :, :2], g_c['u_data'][:, :, :2])
        assert not np.array_equal(g_w['u_data'][:, :, 2:], g_c['u_data'][:, :, 2:])


def test_mpc_planner_takes_nn_config_dir_like_the_reference(tmp_path):
    """evaluate.py:191: MPC_Planner(..., use_NN_cost2go=True, nn_config_dir=get_scenario_config(sc)) -- the reference's
    own gt_mpc call shape -- works unchanged: the YAML names the scenario's checkpoint, the shipped weights are taken,
    identity statistics are announced by a warning, and the planner answers what a planner given the same network
    explicitly answers."""
    import igtmpc
    cfg = tmp_path / 'game_theoretic_NN' / 'configs'
    cfg.mkdir(parents=True)
    (cfg / 'sc1_config.yaml').write_text('model_path: /game_theoretic_NN/models/V_GT_sc1.pt\ninclude_route: False\n'
                                         'hidden_size: 128\nnum_layers: 2\ninput_size: 6\n')
    routes, agents, refs = _scene()
    N = 20
    pred = igtmpc.ConstantAccelerationModel(N=N, dt=0.1)
    preds = pred.predict(agents, [igtmpc.VehicleAction({'a': 0.0, 'df': 0.0}) for _ in routes], routes, refs)
    kw = dict(N=N, dt=0.1, ca_radius=2.8, agents=agents, routes=routes, ref=refs, goals=None, road_dim=(11.4, 50),
              ds_right=8.6, index=0, num_rk4_steps=4, use_NN_cost2go=True, ca_type='circle', weights=[1, 1, 1])
    with pytest.warns(UserWarning, match='normalisation statistics'):
        a = igtmpc.MPC_Planner(nn_config_dir=str(cfg / 'sc1_config.yaml'), **kw)
    b = igtmpc.MPC_Planner(value_net=igtmpc.shipped_value_net(1), **kw)
    assert a._solver is b._solver                                   # same content -> same shared handle
    out = []
    for pl in (a, b):
        pl.update_initial_condition(agents[0], igtmpc.VehicleAction({'a': 0.0, 'df': 0.0}))
        pl.update_predictions(preds, raw_preds=preds)
        out.append(pl.solve())
    assert out[0][2] and out[1][2]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    with pytest.warns(UserWarning):                                  # no file there: the path's own name says which scenario
        c = igtmpc.MPC_Planner(nn_config_dir='/somewhere/else/game_theoretic_NN/configs/sc1_config.yaml', **kw)
    assert c._solver is a._solver
    with pytest.raises(ValueError):
        igtmpc.MPC_Planner(nn_config_dir=str(tmp_path / 'nothing.yaml'), **kw)


def test_c_abi_allgather_controls_single_rank():
    """igt_comm_* / igt_allgather_controls_* through ctypes on one GPU: without a communicator the gather is the strided
    copy u*[:, :, 0]; with a world-of-one RCCL communicator (same ncclAllGather call every rank of an N-GPU job makes)
    the result is the same.  The N > 1 exchange itself is covered by the gloo tests and bench.py --gpus N."""
    import ctypes as ct
    import torch
    import igtmpc
    from igtmpc import _lib as L
    from igtmpc.scenarios import make_batch
    b = make_batch(4096, dtype=np.float64)
    dev = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
           for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
    with igtmpc.BatchSolver(dtype='f64') as s:
        out = s.solve(*dev)
        want = out['u'][:, :, 0].contiguous()
        got = torch.full((4096, 2), -7.0, dtype=torch.float64, device='cuda')
        st = torch.cuda.current_stream().cuda_stream or 1
        L.check(s.lib.igt_allgather_controls_f64(s._h, 4096, out['u'].data_ptr(), got.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(got.nan_to_num(), want.nan_to_num())
        uid = (ct.c_char * 128)()
        L.check(s.lib.igt_comm_unique_id(uid))
        L.check(s.lib.igt_comm_init(s._h, 1, 0, uid))
        with pytest.raises(igtmpc.IgtError):
            L.check(s.lib.igt_comm_init(s._h, 1, 0, uid))          # already initialised
        got2 = torch.full((4096, 2), -7.0, dtype=torch.float64, device='cuda')
        L.check(s.lib.igt_allgather_controls_f64(s._h, 4096, out['u'].data_ptr(), got2.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(got2.nan_to_num(), want.nan_to_num())
        # with a communicator an empty shard, or a shard size other than the first call's, is refused -- a rank that
        # skipped the collective would leave its peers waiting inside ncclAllGather
        for B_bad in (0, 2048):
            with pytest.raises(igtmpc.IgtError):
                L.check(s.lib.igt_allgather_controls_f64(s._h, B_bad, out['u'].data_ptr(), got2.data_ptr(), st))
        L.check(s.lib.igt_comm_destroy(s._h))
        L.check(s.lib.igt_allgather_controls_f64(s._h, 0, out['u'].data_ptr(), got2.data_ptr(), st))     # no communicator: a no-op
        L.check(s.lib.igt_allgather_controls_f64(s._h, 2048, out['u'].data_ptr(), got2.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(got2[:2048].nan_to_num(), want[:2048].nan_to_num())


def test_frenet_step_f32_outside_the_speed_box():
    """igt_frenet_step_f32 has no verdicts: a caller may step states far outside the planner's speed box (the predictor
    allows v up to 20 m/s).  It must stay within 1e-5 of the float64 step there too."""
    import igtmpc
    rng = np.random.default_rng(3)
    n = 512
    x = np.column_stack([rng.uniform(0, 50, n), rng.uniform(-20, 30, n), rng.uniform(5, 40, n), rng.uniform(-0.2, 0.2, n),
                         rng.uniform(-0.2, 0.2, n), rng.uniform(15, 20, n), rng.uniform(-3, 3, n)])
    u = np.column_stack([rng.uniform(-4, 3, n), rng.uniform(-0.6, 0.6, n)])
    kp = np.tile(np.array([19.3, 19.3 + 8.6 * np.pi / 2, 1 / 8.6]), (n, 1))
    x32, u32, kp32 = x.astype(np.float32), u.astype(np.float32), kp.astype(np.float32)
    with igtmpc.BatchSolver(dtype='f32', N=1, C=64, n_obs=0) as s:
        got = s.frenet_step(x32, u32, kp32)
    ref = O.frenet_rk4_step(x32.astype(np.float64), u32[:, 0].astype(np.float64), u32[:, 1].astype(np.float64),
                            kp32.astype(np.float64), O.Params())
    far = np.minimum(np.abs(ref[:, 2] - kp[:, 0]), np.abs(ref[:, 2] - kp[:, 1])) > 2.5     # a step travels up to 2 m
    far &= np.minimum(np.abs(x[:, 2] - kp[:, 0]), np.abs(x[:, 2] - kp[:, 1])) > 2.5
    assert far.sum() > 100 and rel_err(got[far], ref[far]).max() < REL_TOL


def test_plain_c_caller_gets_what_python_gets(tmp_path):
    """examples/c_caller.c (C99, host buffers, igt_solve_batch_f64 through include/igtmpc.h alone) on its built-in batch
    of 256 vehicles on route '12': the same problems through the Python face must give the same winner, cost and first
    control as the C program prints."""
    import re
    import subprocess
    import igtmpc
    from test_host_logic import _build_c_caller
    exe = _build_c_caller(tmp_path / 'c_caller')
    r = subprocess.run([str(exe), '256'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    m = re.match(r'solved (\d+) of 256; scenario (\d+): candidate (\d+), cost (\S+), applies a = (\S+) df = (\S+), s_N = (\S+)', r.stdout)
    assert m, r.stdout
    solved, first, cand, cost, a0, df0, sN = int(m[1]), int(m[2]), int(m[3]), float(m[4]), float(m[5]), float(m[6]), float(m[7])
    B, N = 256, 20
    b = np.arange(B)
    s = 5.0 + 30.0 * b / B
    v = 1.0 + 3.0 * ((b * 7) % B) / B
    x0 = np.column_stack([s, np.full(B, 2.8), s, np.full(B, 0.01), np.full(B, -0.005), v, np.zeros(B)])
    kp = np.tile([19.3, 19.3 + 8.6 * np.arccos(-1.0) / 2, 1.0 / 8.6], (B, 1))
    obs = np.full((B, 1, 2, N + 1), -20.0)
    with igtmpc.BatchSolver(dtype='f64') as sv:
        out = sv.solve(x0, np.tile([0.1, 0.0], (B, 1)), kp, np.zeros(B, np.uint32), obs)
    assert int((out['status'] == 0).sum()) == solved > 100
    assert int(np.nonzero(out['status'] == 0)[0][0]) == first and int(out['argmin'][first]) == cand
    assert abs(out['cost'][first] - cost) < 1e-5 and abs(out['u'][first, 0, 0] - a0) < 1e-4 and abs(out['u'][first, 1, 0] - df0) < 1e-4
    assert abs(out['x'][first, 2, N] - sN) < 1e-4


@pytest.mark.parametrize('mode', ['mpc', 'gt_mpc'])
def test_driver_command_line_writes_the_reference_run_directory(tmp_path, mode):
    """`python -m igtmpc.evaluate --save_dir D/ --eval_mode M --sc 1 --num_samples 3` -- the reference's command line
    (evaluate.py:643-650) -- leaves the reference driver's files: [episodes, 7 M, T+1] / [episodes, 2 M, T] arrays of a 15 s
    run, one csv row per episode."""
    import csv
    import json
    import pickle
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, 'igt-mpc-int_amd'))
    out = subprocess.run([sys.executable, '-m', 'igtmpc.evaluate', '--save_dir', str(tmp_path) + '/', '--eval_mode', mode, '--sc', '1',
                          '--num_samples', '3', '--N', '20'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    run = line['run_dir']
    assert os.path.basename(run).startswith(f'{mode}_sc1_seed2026_')
    sub = run + ('/game_mpc/evaluation' if mode == 'gt_mpc' else '/mpc')
    with open(sub + '/cl_traj.pkl', 'rb') as f:      # written by this test's own child process a moment ago
        cl = pickle.load(f)
    with open(sub + '/u_cl.pkl', 'rb') as f:
        ucl = pickle.load(f)
    assert cl.shape == (3, 14, 151) and ucl.shape == (3, 4, 150) and np.isfinite(cl).all() and np.isfinite(ucl).all()
    assert (cl[:, 2::7, -1] >= cl[:, 2::7, 0]).all() and (cl[:, 5::7, 0] == 0).all()    # a standing start (v0 = 0), nobody rolled back
    assert mode == 'gt_mpc' or (cl[:, 2::7, -1] > cl[:, 2::7, 0] + 15).all()          # everybody made way (the shipped value nets run without their statistics)
    with open(sub + ('/stats.csv' if mode == 'gt_mpc' else '/eval_stats.csv'), newline='') as f:
        rows = list(csv.DictReader(f))
    assert len(rows) == 3 and ('NN_query_time' in rows[0]) == (mode == 'gt_mpc')


def test_closed_loop_with_the_policy_files_other_settings_matches_oracle_loop():
    """What load_reference_configs hands on from mpc.yaml / fourwayint.yaml other than the shipped values: constant-speed
    forecasts (mpc.yaml:14), a gentler brake fallback (a_min), a rolling start (v0) -- host loop and device-resident loop
    against the oracle's loop with the same settings, 1e-9."""
    import closed_loop as CL
    from igtmpc import routes as R
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.evaluate import initial_states, run_closed_loop
    pairs = [R.SCENARIO_ROUTES[0][0], R.SCENARIO_ROUTES[5][2], ('13', '23')]
    x, _ = initial_states(np.random.default_rng(7), pairs, v0=2.0)
    assert (x[:, :, 5] == 2.0).all()
    # third episode: agent 0 outside the lane bound (brake fallback from step 0, so its forecast is never replaced by a shared
    # plan) and fast enough that "holds 4 m/s" and "brakes at 2.5 m/s^2" put it on different sides of agent 1's path
    for m, (r, s0, v0) in enumerate(zip(pairs[2], (17.0, 12.0), (4.0, 3.0))):
        xy = R.frenet2global(R.ROUTE_ID[r], s0)
        x[2, m] = (xy[0], xy[1], s0, 0.0, 0.0, v0, float(R.psi_ref(R.ROUTE_ID[r], s0)))
    x[2, 0, 3] = 0.25
    P = O.Params(N=20)
    kw = dict(N=20, T_sim=2.5, dtype='f64', cand_mode='track', init=(x, pairs), a_min_policy=-2.5, constant_speed=True, warm_start=True)
    got = run_closed_loop(**kw)
    dev = run_closed_loop(device_resident=True, **kw)
    plain = run_closed_loop(**dict(kw, constant_speed=False))
    assert not np.array_equal(got['x_data'], plain['x_data'])          # the switch reaches the forecast
    fb = 0
    for e in range(len(pairs)):
        ref = CL.run_episode(x[e], pairs[e], P, cinf_halfplanes(), M_sim=25, cand_mode='track', track_env=got['track_env'],
                             a_min_policy=-2.5, constant_speed=True)
        fb += ref['events']['fallback']
        for r in (got, dev):
            assert rel_err(r['x_data'][e], ref['x_data']).max() < 1e-9, (e, pairs[e])
            assert rel_err(r['u_data'][e], ref['u_data']).max() < 1e-9, (e, pairs[e])
    assert fb > 0
    assert (got['u_data'][2, 0, :3] == -2.5).all()                    # the fallback brakes with the policy file's a_min
