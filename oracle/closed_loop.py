"""
TEST INFRASTRUCTURE ONLY -- float64 restatement of the reference's closed-loop time loop.

Restates evaluate.py:451-569 (--eval_mode mpc) and evaluate.py:202-330 (--eval_mode gt_mpc) for ONE episode at a time,
agent by agent, in plain Python with the reference's own variable roles, so that igtmpc/evaluate.py (E episodes in
lock-step, batched, on the GPU) has something independent to be compared with:

    for t in range(M_sim):                                                        evaluate.py:451
        preds      = predictor.predict(cur states, prev inputs)                   :455   (np_oracle.predict_constant_accel)
        preds4CAV  = share_motion_forecasts(preds, cav_sols of step t-1)          :458-460, utils.py:339-352
        for i in range(M):                                                        :469   (Jacobi: all agents see preds4CAV)
            update_initial_condition(state_i, u_prev_i)                           :470
            update_predictions(filter_preds(preds4CAV, i))                        :474-477, utils.py:365-388
            warm start = augment_prev_sol(solution of step t-1) if i solved then  :478-481, utils.py:354-363
            x_sol, u_sol, ok = solve(warm start)                                  :482   (np_oracle shooting solve)
            ok:   next state = x_sol[:,1], applied input = u_sol[:,0]             :491-510
            else: brake a = a_min if v > 0 else 0, keep df, one model step;       :511-545
                  v < 0: applied a = 0, state frozen with v = 0                   :523-526
    deadlock = at least two agents end with s <= 30                               :566-569

What the gt_mpc branch does differently (eval_mode='gt_mpc'):
    previous inputs start at (0, 0), not (0.1, 0)                                 evaluate.py:171 vs 419
    t == 0: the forecast is made with a = 0.09 (k + 1) for agent k, df = 0        :207-210
    the warm start is passed for t > 1 only (mpc mode: t >= 1)                    :232 vs 478
    cost: - V(W (x_N - mu)) sigma_t + mu_t replaces - (s_N - s_0)                 mpc.py:367-369
    x_N = [s_tv, v_tv, e_tv, s_N - s_tv, v_N - v_tv, e_ego - e_tv] with (s_tv, v_tv) the LAST state of the other agent's
    forecast as shared but NOT filtered (raw_preds = preds4CAV, evaluate.py:229; mpc.py:330) and e the scenario encoding
    (utils.py:141-169; tests/golden/scenario_encoding.json holds the reference's own answers)

What stands in for IPOPT is the sampled shooting solve of np_oracle (same candidate families as the device), so this
pins the LOOP semantics -- who sees which forecast when, what the fallback does, what is shared -- not IPOPT's optimum.
Parity status: PARITY UNPINNED against the reference itself (evaluate.py needs casadi / IPOPT / polytope, which are
not installable here); restated from source with line citations.
"""
import json
import os

import numpy as np

import np_oracle as O

A_MIN_POLICY = -4.0       # mpc.yaml:8 `a_min`, used by the brake fallback (evaluate.py:514)
ABS_HEADING_ROUTES = ('32', '41')       # mpc.py:231, 282

_CONST = None


def route_constants():
    """tests/golden/route_constants.json -- produced by running the reference's ReferenceGen.py (make_golden.py)."""
    global _CONST
    if _CONST is None:
        here = os.path.dirname(os.path.abspath(__file__))
        with open(os.path.join(here, '..', 'tests', 'golden', 'route_constants.json')) as f:
            _CONST = json.load(f)
    return _CONST


def kparams_of(route):
    """Curvature function of the route as (b0, b1, Kv)  (mpc.py:183-200; straight: K == 0)."""
    k = route_constants()[route]
    return np.array([np.inf, np.inf, 0.0]) if k['straight'] else np.array([k['b0'], k['b1'], k['Kv']])


_ENC = None


def scenario_encoding(routes):
    """utils.py:141-169 -- looked up in the fixture the reference's own function produced (make_golden.py)."""
    global _ENC
    if _ENC is None:
        here = os.path.dirname(os.path.abspath(__file__))
        with open(os.path.join(here, '..', 'tests', 'golden', 'scenario_encoding.json')) as f:
            _ENC = json.load(f)['scenario_encoding']
    e = _ENC[f'{routes[0]},{routes[1]}']
    if e is None:
        raise ValueError('Scenario not found')
    return e


def augment_prev_sol(x_sol_prev, u_sol_prev, kp, P):
    """utils.py:354-363: shift the previous solution by one step; extend the states by one model step with the last
    control (retried with a = 0 if that step ends above v = 5; v clipped to [-1, 5]) and the controls by repeating the
    last one.  -> (x[7,N+1], u[2,N])."""
    last = x_sol_prev[:, -1]
    nxt = O.frenet_rk4_step(last, u_sol_prev[0, -1], u_sol_prev[1, -1], kp, P)
    if nxt[5] > 5:                                                              # utils.py:359-360
        nxt = O.frenet_rk4_step(last, 0.0, u_sol_prev[1, -1], kp, P)
    nxt = nxt.copy()
    nxt[5] = np.clip(nxt[5], -1, 5)                                             # utils.py:361
    x = np.hstack([x_sol_prev[:, 1:], nxt[:, None]])
    u = np.hstack([u_sol_prev[:, 1:], u_sol_prev[:, [-1]]])                     # utils.py:362
    return x, u


def run_episode(x_init, routes, P, cinf, M_sim=30, cand_mode='lattice', C=256, refine_iters=0, warm_start=True,
                u_init=None, eval_mode='mpc', net=None, track_env=1.0, track_vcap=1.0, a_min_policy=A_MIN_POLICY, constant_speed=False):
    """x_init[M,7] (planner state order), routes = (route of agent 0, route of agent 1).
    eval_mode 'gt_mpc' needs net = dict(layers, Wn, mu_f, sigma_t, mu_t) (np_oracle.terminal_value).
    track_env: scale of the tracking family's acceleration envelope (the driver under test picks it from the horizon).
    a_min_policy: mpc.yaml:8 a_min of the brake fallback (evaluate.py:514); constant_speed: the other agent is forecast with
    a = 0 (mpc.yaml:13-14 prediction_type, evaluate.py:78-79, constant_acceleration_model.py:26-29).
    -> dict(x_data[7M, M_sim+1], u_data[2M, M_sim], infeasible[M], deadlock, events)."""
    M, N, dt = len(routes), P.N, P.dt
    assert M == 2
    gt = eval_mode == 'gt_mpc'
    consts = route_constants()
    A, b = cinf
    kp = [kparams_of(r) for r in routes]
    cur = [np.array(x_init[i], dtype=np.float64) for i in range(M)]
    if u_init is None:
        u_init = (0.0, 0.0) if gt else (0.1, 0.0)                                # evaluate.py:171 / 419
    code = scenario_encoding(routes) if gt else None                             # mpc.py:336-337
    prev_in = [np.array(u_init, dtype=np.float64) for _ in range(M)]
    x_data = np.zeros((7 * M, M_sim + 1))
    u_data = np.zeros((2 * M, M_sim))
    for i in range(M):
        x_data[7 * i:7 * i + 7, 0] = cur[i]
    prev_sols, prev_idx = [], []                                                # evaluate.py:444-445
    infeasible = np.zeros(M, dtype=np.int64)
    events = dict(fallback=0, stop=0, share=0, share_retry=0, warm=0)
    for t in range(M_sim):
        nxt_states, cur_in, sols, idx = [None] * M, [None] * M, [], []
        for i in range(M):
            j = 1 - i
            plan = prev_sols[prev_idx.index(j)] if (t > 0 and j in prev_idx) else None     # evaluate.py:459
            if plan is not None:
                events['share'] += 1
                if np.clip(plan[0][5, N] + plan[1][0, N - 1] * dt, -2.0, 20.0) > 5:        # utils.py:348 (inside forecast_for_ego)
                    events['share_retry'] += 1
            a_fc = prev_in[j][0]
            if gt and t == 0:
                a_fc = 0.0 + 0.09 * (j + 1)                                                # evaluate.py:207-210
            if constant_speed:
                a_fc = 0.0
            obs, tv = O.forecast_for_ego(routes[j], consts[routes[j]], cur[i][:2], cur[i][6], cur[j], a_fc, N, dt,
                                         None if plan is None else plan[0], None if plan is None else plan[1])
            flags = np.array([O.FLAG_ABS_HEADING if routes[i] in ABS_HEADING_ROUTES else 0], dtype=np.uint32)
            u_ws = None
            if warm_start and cand_mode in ('ramp_hold', 'track') and i in prev_idx and t > (1 if gt else 0):   # :478 / :232
                xs_p, us_p = prev_sols[prev_idx.index(i)]
                u_ws = augment_prev_sol(xs_p, us_p, kp[i], P)[1][None]
                flags = flags | np.uint32(O.FLAG_WARM)
                events['warm'] += 1
            args = (cur[i][None], prev_in[i][None], kp[i][None], flags, obs[None, None], A, b, P)
            kw = {}
            if gt:
                kw = dict(net=net, tv_sv=np.array([tv], dtype=np.float64),
                          enc=np.array([[code[i], code[j]]], dtype=np.float64))
            if cand_mode in ('ramp_hold', 'track'):
                r = O.solve_batch_refined(*args, C=C, refine_iters=refine_iters, u_ws=u_ws, cand=cand_mode,
                                          track=dict(env=track_env, vcap=track_vcap), **kw)[-1]
            else:
                r = O.solve_batch(*args, C=C, **kw)
            if r['status'][0] == 0:                                                        # evaluate.py:484-510
                xs, us = r['x'][0], r['u'][0]
                sols.append((xs, us))
                idx.append(i)
                nxt_states[i] = xs[:, 1].copy()
                cur_in[i] = us[:, 0].copy()
            else:                                                                          # evaluate.py:511-545
                infeasible[i] += 1
                events['fallback'] += 1
                a_fb = a_min_policy if cur[i][5] > 0 else 0.0
                df_fb = prev_in[i][1]
                ns = O.frenet_rk4_step(cur[i], a_fb, df_fb, kp[i], P)
                if cur[i][5] < 0:                                                          # evaluate.py:523-526
                    events['stop'] += 1
                    cur_in[i] = np.array([0.0, df_fb])
                    ns = cur[i].copy()
                    ns[5] = 0.0
                else:
                    cur_in[i] = np.array([a_fb, df_fb])
                nxt_states[i] = ns
            x_data[7 * i:7 * i + 7, t + 1] = nxt_states[i]
            u_data[2 * i:2 * i + 2, t] = cur_in[i]
        cur, prev_in = nxt_states, cur_in                                                  # evaluate.py:556-557
        prev_sols, prev_idx = sols, idx                                                    # evaluate.py:560-561
    deadlock = bool(sum(x_data[7 * i + 2, -1] <= 30 for i in range(M)) >= 2)              # evaluate.py:566-569
    return dict(x_data=x_data, u_data=u_data, infeasible=infeasible, deadlock=deadlock, events=events)
