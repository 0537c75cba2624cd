"""Closed-loop driver: the time loop of the reference's evaluate.py (mpc branch 451-564, gt_mpc
branch 193-330) for E episodes in lock-step, every (episode, agent) problem of a timestep solved
in ONE batched call on the GPU (the reference's agents of a timestep all read the same
predictions -- a Jacobi update, evaluate.py:469-558 -- so batching changes nothing).

Kept from the reference, with citations:
  * initial states: random offset along the approach lane, v0 = 0, s = offset      (evaluate.py:91-94, 404-418)
  * previous inputs start at (0.1, 0)                                               (evaluate.py:419)
  * gt_mpc: previous inputs start at (0, 0) (:171); the first forecast uses a = 0.09 (k+1) for agent k (:207-210);
    the warm start is passed for t > 1 only (:232; the mpc branch passes it from t = 1, :478)
  * per step: predict -> share motion forecasts (V2V) -> filter_preds -> solve      (evaluate.py:455-482)
  * solved: next state = x*[:,1], applied input = u*[:,0]                           (evaluate.py:491-510)
  * infeasible: brake a = a_min (if v > 0 else 0), keep df, one model step; v < 0 -> stop  (evaluate.py:511-545)
  * deadlock: at least two agents end with s <= 30                                  (evaluate.py:566-569)
Not kept: pickles / CSV / mp4 output, the unseeded route shuffle (routes are sorted pairs here).

    python -m igtmpc.evaluate --sc 1 --num_samples 8 --N 20
"""
import argparse
import json
import os
import time

import numpy as np

from . import routes as R
from .cinf import cinf_halfplanes
from ._lib import IGT_FLAG_WARM
from .solver import BatchSolver
from .value_nets import shipped_value_net

A_MIN_POLICY = -4.0        # mpc.yaml:8, used by the brake fallback (evaluate.py:514)


def initial_states(rng, route_pairs, v0=0.0):
    """evaluate.py:91-94 + 404-418: every episode draws one offset per approach lane (order 1,2,3,4);
    an agent starting at origin o sits `offset_o` metres down its lane with v = 0, ey = epsi = 0."""
    E = len(route_pairs)
    max_start = (R.ROAD_LENGTH - R.ROAD_WIDTH) / 2 - (R.ROAD_WIDTH - R.CA_RADIUS)      # evaluate.py:48-49
    off = rng.random((E, 4)) * max_start
    M = 2
    x = np.zeros((E, M, 7))
    rid = np.zeros((E, M), dtype=np.int64)
    for e, pair in enumerate(route_pairs):
        for m, r in enumerate(pair):
            rid[e, m] = R.ROUTE_ID[r]
            s0 = off[e, int(r[0]) - 1]
            xy = R.frenet2global(rid[e, m], s0)
            h0 = {'1': 0.0, '2': -np.pi / 2, '3': -np.pi, '4': np.pi / 2}[r[0]]     # evaluate.py:58-61
            x[e, m] = (xy[0], xy[1], s0, 0.0, 0.0, v0, h0)                          # fourwayint.yaml:11 v0
    return x, rid


def auto_track_env(N, dt):
    """Scale of the tracking family's acceleration envelope (igt_params.track_env) for a CLOSED loop with horizon N dt.
    1 is the derived value -- the stationary acceleration of the NLP's own cost -- and the best setting from a 4 s horizon
    on (mpc.yaml:6 ships N = 40, dt = 0.1); a plan that is optimal over a shorter horizon runs faster into conflicts it
    cannot see yet, so the scale shrinks with the horizon, not below 0.25.  Measured (tools/envelope_sweep.py, 512 episodes):
    N = 20: 0.5 -> 9.2 % infeasible steps / 23 % deadlock flag / 47.6 m against 14.9 % / 29 % / 46.4 m at 1.0;
    N = 40: 1.0 -> 3.3 % / 2.9 % / 60.1 m (0.5: 3.0 % / 2.5 % / 61.1 m) -- round 4, speed cap on, no warm start.
    The library's own default stays 1.0 (one solve knows no closed loop)."""
    return float(min(1.0, max(0.25, N * dt / 4.0)))


def run_closed_loop(sc=1, num_samples=1, N=40, dt=0.1, T_sim=15.0, seed=2026, C=256, n_rk4=4, device=0,
                    dtype='f64', rotation=None, cand_mode='track', refine_iters=0, verbose=False,
                    eval_mode='mpc', value_net=None, device_resident=False, warm_start=None, init=None,
                    terminal_set=True, feas_tol=None, limits=None, graph=False, a_min_policy=None, constant_speed=False,
                    v0=0.0):
    """eval_mode 'mpc' (evaluate.py:370-639) or 'gt_mpc' (123-369: terminal value network in the cost;
    value_net = dict(layers=[(W,b),...][, Wn, mu_f, sigma_t, mu_t]), default: the network the reference ships for
    scenario sc -- its normalisation statistics are not shipped, identity unless given).  device_resident=True keeps every per-step array in HBM (torch tensors;
    forecast, solve, fallback step and the state update never leave the GPU) -- for thousands of episodes.
    warm_start (ramp-hold and tracking candidates): an agent that solved the previous step centres its candidates on that
    solution shifted by one step (evaluate.py:478-481, utils.py:354-363 augment_prev_sol) instead of on u_prev held.  Default
    (None): on for ramp-hold, OFF for the tracking family -- with the speed cap of its targets the loop is better without on
    every count at both horizons (N = 40: 3.3 % infeasible steps / 2.9 % deadlock flag / 60.1 m against 5.0 % / 3.9 % /
    57.2 m; N = 20: 9.2 % / 23 % / 47.6 m against 9.7 % / 29 % / 46.2 m).  The reference's warm start
    is IPOPT's initial guess (mpc.py:386-389): it does not move the NLP's optimum, so nothing of its semantics is lost.
    init = (x[E,M,7], route_pairs[E]) overrides the sampled initial states; terminal_set=False drops the C_inf
    constraint (mpc.py:177-180) -- a test switch.  feas_tol: inequality tolerance of the verdicts (default: the
    library's 1e-6; IPOPT's constr_viol_tol is 1e-3, mpc.py:135); limits: further igt_params fields by name
    (e.g. dict(track_env=0.0); without it the tracking family's envelope scale follows the horizon: auto_track_env).
    graph=True (with device_resident): the time loop replays one captured step.
    a_min_policy: deceleration of the brake fallback (mpc.yaml:8 a_min through evaluate.py:514; default -4);
    constant_speed: the forecast holds the other agent's speed (mpc.yaml:13-14 prediction_type, evaluate.py:76-79);
    v0: initial speed (fourwayint.yaml:11) -- load_reference_configs reads all three from the reference's files."""
    a_min_policy = A_MIN_POLICY if a_min_policy is None else float(a_min_policy)
    gt = eval_mode == 'gt_mpc'
    if gt and value_net is None:
        if init is not None:
            raise ValueError("eval_mode='gt_mpc' with init= needs value_net (the shipped networks are per scenario)")
        value_net = shipped_value_net(sc)                               # sc{n}_config.yaml:2 model_path
    rng = np.random.default_rng(seed)                                   # evaluate.py:35, 56
    M = 2
    M_sim = int(round(T_sim / dt))                                      # evaluate.py:83-84
    if init is not None:
        x, pairs = np.array(init[0], dtype=np.float64), [tuple(p) for p in init[1]]
        E = num_samples = len(pairs)
        rid = np.array([[R.ROUTE_ID[r] for r in p] for p in pairs], dtype=np.int64)
    else:
        E = num_samples
        pairs = [R.SCENARIO_ROUTES[sc - 1][(e if rotation is None else rotation) % 4] for e in range(E)]
        x, rid = initial_states(rng, pairs, v0=v0)                      # x[E,M,7]
    warm = (cand_mode == 'ramp_hold') if warm_start is None else (bool(warm_start) and cand_mode in ('ramp_hold', 'track'))
    limits = dict(limits or {})
    if cand_mode == 'track' and 'track_env' not in limits:
        limits['track_env'] = auto_track_env(N, dt)                     # (not applied with the gt_mpc cost: igtmpc.h)
    kp = R.kparams(rid)                                                 # [E,M,3]
    absh = R.TABLES['abs_heading'][rid]
    flags = absh.astype(np.uint32).reshape(-1)
    u_prev = np.tile(np.array([0.0 if gt else 0.1, 0.0]), (E, M, 1))    # evaluate.py:419 (mpc) / 171 (gt_mpc)
    solver = BatchSolver(N=N, dt=dt, n_rk4=n_rk4, C=C, n_obs=M - 1, device=device, dtype=dtype, cand_mode=cand_mode,
                         refine_iters=refine_iters if cand_mode in ('ramp_hold', 'track') else 0,
                         cost_mode='value_net' if gt else 'progress',
                         **dict({} if feas_tol is None else {'feas_tol': feas_tol}, **(limits or {})))
    if gt:
        solver.set_value_net(**value_net)
        # scenario encodings (mpc.py:336-337, utils.py:84-169): (e_ego, e_other) per problem
        enc = np.zeros((E, M, 2))
        for e, pair in enumerate(pairs):
            code = R.scenario_encoding_sign(pair, R.scenario_of(pair))
            enc[e, 0] = (code[0], code[1])
            enc[e, 1] = (code[1], code[0])
        enc = enc.reshape(E * M, 2)
    if terminal_set:
        solver.set_cinf(*cinf_halfplanes(dt=dt, jerk=solver.params.jerk_limit))
    stepper = BatchSolver(N=1, dt=dt, n_rk4=n_rk4, C=64, n_obs=0, device=device, dtype='f64',
                          **{k: v for k, v in limits.items() if k in ('l_r', 'l_f')})      # the same vehicle
    npdt = solver.np_dtype

    if device_resident:
        out = _loop_device(solver, stepper, x, u_prev, kp, flags, rid, enc if gt else None, gt, E, M, N, M_sim, device, warm,
                           graph=graph, a_min_policy=a_min_policy, constant_speed=constant_speed)
        solver.close()
        stepper.close()
        out['routes'] = pairs
        out['track_env'] = limits.get('track_env')
        return out

    x_data = np.zeros((E, 7 * M, M_sim + 1))
    u_data = np.zeros((E, 2 * M, M_sim))
    x_data[:, :, 0] = x.reshape(E, 7 * M)
    infeasible = np.zeros((E, M), dtype=np.int64)
    solve_ms = []
    have_sol = np.zeros((E, M), dtype=bool)
    sol_x = np.zeros((E, M, 7, N + 1))
    sol_u = np.zeros((E, M, 2, N))

    for t in range(M_sim):
        # --- forecast of the other agent for every (episode, ego) problem, on the device:
        #     predict (evaluate.py:455) -> share motion forecasts (458-460) -> filter_preds (474)
        other = slice(None, None, -1)
        ego_xyh = x[:, :, [0, 1, 6]].reshape(E * M, 3)
        opp = x[:, other][:, :, [0, 1, 2, 5]].reshape(E * M, 4)
        a_fc = u_prev[:, other, 0]
        if gt and t == 0:                                               # evaluate.py:207-210: a = 0.09 (k+1) for agent k
            a_fc = np.tile(0.09 * (np.arange(M)[::-1] + 1.0), (E, 1))
        if constant_speed:                                              # constant_acceleration_model.py:26-29
            a_fc = np.zeros((E, M))
        obs, tv = solver.forecast(ego_xyh.astype(npdt), opp.astype(npdt), a_fc.reshape(-1).astype(npdt),
                                   rid[:, other].reshape(-1).astype(np.int32),
                                   sol_x[:, other].reshape(E * M, 7, N + 1).astype(npdt),
                                   sol_u[:, other].reshape(E * M, 2, N).astype(npdt),
                                   (have_sol[:, other] & (t > 0)).reshape(-1).astype(np.int32))
        # --- solve every (episode, agent) problem at once (evaluate.py:470-482); an agent that solved the previous
        #     step passes that solution, shifted by one step, as its warm start (evaluate.py:478-481)
        t0 = time.perf_counter()
        fl, u_ws = flags, None
        if warm and t > (1 if gt else 0):                              # evaluate.py:478 (mpc: t >= 1) / :232 (gt_mpc: t > 1)
            fl = flags | np.where(have_sol.reshape(-1), IGT_FLAG_WARM, 0).astype(np.uint32)
            u_ws = shift_controls(sol_u).reshape(E * M, 2, N).astype(npdt)
        out = solver.solve(x.reshape(E * M, 7).astype(npdt), u_prev.reshape(E * M, 2).astype(npdt),
                           kp.reshape(E * M, 3).astype(npdt), fl, obs,
                           tv if gt else None, enc.astype(npdt) if gt else None, u_ws=u_ws)
        solve_ms.append((time.perf_counter() - t0) * 1e3)
        ok = (out['status'] == 0).reshape(E, M)
        xs = out['x'].reshape(E, M, 7, N + 1).astype(np.float64)
        us = out['u'].reshape(E, M, 2, N).astype(np.float64)
        # --- infeasible: brake fallback (evaluate.py:511-545)
        v_now = x[..., 5]
        a_fb = np.where(v_now > 0, a_min_policy, 0.0)
        u_fb = np.stack([a_fb, u_prev[..., 1]], axis=-1)
        nxt_fb = stepper.frenet_step(x.reshape(E * M, 7), u_fb.reshape(E * M, 2), kp.reshape(E * M, 3)).reshape(E, M, 7)
        neg = v_now < 0                                                 # evaluate.py:523-526: instantaneous stop
        stop = x.copy()
        stop[..., 5] = 0.0
        nxt_fb = np.where(neg[..., None], stop, nxt_fb)
        u_fb[..., 0] = np.where(neg, 0.0, u_fb[..., 0])
        x_next = np.where(ok[..., None], xs[:, :, :, 1], nxt_fb)
        u_app = np.where(ok[..., None], us[:, :, :, 0], u_fb)
        infeasible += ~ok
        x, u_prev = x_next, u_app
        have_sol, sol_x, sol_u = ok, np.nan_to_num(xs), np.nan_to_num(us)
        x_data[:, :, t + 1] = x.reshape(E, 7 * M)
        u_data[:, :, t] = u_app.reshape(E, 2 * M)
        if verbose and t % 10 == 0:
            print(f't={t:3d} s={x[0, :, 2].round(2)} v={x[0, :, 5].round(2)} ok={ok[0]}', flush=True)

    deadlock = (x_data[:, 2::7, -1] <= 30).sum(axis=1) >= 2             # evaluate.py:566-569
    solver.close()
    stepper.close()
    return dict(x_data=x_data, u_data=u_data, infeasible_ratio=infeasible / M_sim, deadlock=deadlock,
                routes=pairs, solve_ms=np.array(solve_ms), track_env=limits.get('track_env'))


def shift_controls(u):
    """Control part of utils.py:354-363 augment_prev_sol: u[...,2,N] -> [u[..., 1:], u[..., -1:]] (numpy or torch)."""
    if isinstance(u, np.ndarray):
        return np.concatenate([u[..., 1:], u[..., -1:]], axis=-1)
    import torch
    return torch.cat([u[..., 1:], u[..., -1:]], dim=-1)


def _loop_device(solver, stepper, x, u_prev, kp, flags, rid, enc, gt, E, M, N, M_sim, device, warm, graph=False,
                 a_min_policy=A_MIN_POLICY, constant_speed=False):
    """The same time loop with every array resident in HBM (float64 state, solver-dtype views per call).
    graph=True: steps 0 and 1 run eagerly, then ONE step -- forecast, solve, fallback step and the ~40 small tensor
    operations between them -- is captured in a stream graph and replayed for the remaining steps (the state lives in
    fixed buffers that the captured step updates in place; the column of x_data / u_data it writes is a device-side
    counter).  The loop is launch-bound at a few hundred problems per step; the replay removes the launches."""
    import torch
    dev = torch.device('cuda', device)
    td = torch.float32 if solver.dtype == 'f32' else torch.float64
    T = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    x, u_prev, kp = T(x).clone(), T(u_prev).clone(), T(kp)
    kp_s = kp.reshape(E * M, 3).to(td).contiguous()
    kp_d = kp.reshape(E * M, 3).contiguous()
    flags_t = torch.as_tensor(flags.astype(np.int32), device=dev)
    rid_o = torch.as_tensor(rid[:, ::-1].reshape(-1).astype(np.int32), device=dev)
    enc_t = T(enc, td) if gt else None
    x_data = torch.zeros((E, 7 * M, M_sim + 1), dtype=torch.float64, device=dev)
    u_data = torch.zeros((E, 2 * M, M_sim), dtype=torch.float64, device=dev)
    x_data[:, :, 0] = x.reshape(E, 7 * M)
    infeasible = torch.zeros((E, M), dtype=torch.int64, device=dev)
    have_sol = torch.zeros((E, M), dtype=torch.bool, device=dev)
    sol_x = torch.zeros((E, M, 7, N + 1), dtype=td, device=dev)
    sol_u = torch.zeros((E, M, 2, N), dtype=td, device=dev)
    ix = torch.tensor([0, 1, 6], device=dev)
    io = torch.tensor([0, 1, 2, 5], device=dev)
    col = torch.zeros(1, dtype=torch.int64, device=dev)            # the step the next call of step() computes
    a_fc0 = torch.as_tensor(0.09 * (np.arange(M)[::-1] + 1.0), device=dev).expand(E, M) if gt else None

    def step(first, warm_ok=True):
        xo, uo = x.flip(1), u_prev.flip(1)
        a_fc = a_fc0 if (gt and first) else uo[..., 0]
        if constant_speed:
            a_fc = torch.zeros_like(uo[..., 0])
        hp = (have_sol.flip(1) & (not first)).reshape(-1).to(torch.int32).contiguous()
        obs, tv = solver.forecast(x.index_select(2, ix).reshape(E * M, 3).to(td).contiguous(),
                                  xo.index_select(2, io).reshape(E * M, 4).to(td).contiguous(),
                                  a_fc.reshape(-1).to(td).contiguous(), rid_o,
                                  sol_x.flip(1).reshape(E * M, 7, N + 1).contiguous(),
                                  sol_u.flip(1).reshape(E * M, 2, N).contiguous(), hp)
        fl, u_ws = flags_t, None
        if warm and not first and warm_ok:
            fl = flags_t | (have_sol.reshape(-1).to(torch.int32) * IGT_FLAG_WARM)
            u_ws = shift_controls(sol_u).reshape(E * M, 2, N).contiguous()
        out = solver.solve(x.reshape(E * M, 7).to(td).contiguous(), u_prev.reshape(E * M, 2).to(td).contiguous(), kp_s,
                           fl, obs, tv if gt else None, enc_t, u_ws=u_ws)
        ok = (out['status'] == 0).reshape(E, M)
        xs = out['x'].reshape(E, M, 7, N + 1)
        us = out['u'].reshape(E, M, 2, N)
        v_now = x[..., 5]
        a_fb = torch.where(v_now > 0, torch.full_like(v_now, a_min_policy), torch.zeros_like(v_now))
        neg = v_now < 0
        u_fb = torch.stack([torch.where(neg, torch.zeros_like(a_fb), a_fb), u_prev[..., 1]], dim=-1)
        u_step = torch.stack([a_fb, u_prev[..., 1]], dim=-1)
        nxt_fb = stepper.frenet_step(x.reshape(E * M, 7).contiguous(), u_step.reshape(E * M, 2).contiguous(),
                                     kp_d).reshape(E, M, 7)
        stop = x.clone()
        stop[..., 5] = 0.0
        nxt_fb = torch.where(neg[..., None], stop, nxt_fb)
        x_new = torch.where(ok[..., None], xs[:, :, :, 1].to(torch.float64), nxt_fb)
        u_new = torch.where(ok[..., None], us[:, :, :, 0].to(torch.float64), u_fb)
        x.copy_(x_new)
        u_prev.copy_(u_new)
        infeasible.add_((~ok).to(torch.int64))
        have_sol.copy_(ok)
        sol_x.copy_(torch.nan_to_num(xs))
        sol_u.copy_(torch.nan_to_num(us))
        u_data.index_copy_(2, col, u_prev.reshape(E, 2 * M, 1))
        col.add_(1)
        x_data.index_copy_(2, col, x.reshape(E, 7 * M, 1))

    torch.cuda.synchronize(dev)
    t_start = time.perf_counter()
    step(True)
    t = 1
    use_graph = graph and M_sim > 3
    if gt and not use_graph and M_sim > 1:
        step(False, warm_ok=False)             # evaluate.py:232: the gt_mpc loop passes its warm start for t > 1 only
        t = 2
    if use_graph:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            step(False, warm_ok=not gt)        # step 1, eagerly, on the capture stream (gt_mpc: no warm start yet, :232)
        torch.cuda.current_stream(dev).wait_stream(side)
        t = 2
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            step(False)
        for _ in range(t, M_sim):
            g.replay()
        t = M_sim
    for _ in range(t, M_sim):
        step(False)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t_start
    xd = x_data.cpu().numpy()
    return dict(x_data=xd, u_data=u_data.cpu().numpy(), infeasible_ratio=infeasible.cpu().numpy() / M_sim,
                deadlock=(xd[:, 2::7, -1] <= 30).sum(axis=1) >= 2, solve_ms=np.full(M_sim, wall / M_sim * 1e3),
                wall_s=wall)


def load_reference_configs(policy_yaml=None, env_yaml=None):
    """The reference driver's two configuration files -- mpc.yaml (evaluate.py:32-33) and common/fourwayint.yaml
    (evaluate.py:28-29) -- as keyword arguments of run_closed_loop: -> (kwargs, policy) where `policy` is the parsed
    mpc.yaml (what save_results writes back, evaluate.py:327-329).  What the driver reads of them is honoured
    (N, dt, a_min of the brake fallback, prediction_type, v0, ca_radius, l_r, l_f); what this package holds as frozen,
    reference-generated tables is CHECKED and refused when it differs (road geometry and fillet radius behind
    igtmpc/data/route_constants.json, the forecast's speed clip, two agents, the circle constraint) -- a silent mismatch
    would be a different intersection."""
    import yaml
    kw, policy = {}, {}
    if policy_yaml is not None:
        with open(policy_yaml) as f:
            policy = yaml.safe_load(f) or {}
        if policy.get('type', 'MPC') != 'MPC':
            raise ValueError(f"{policy_yaml}: type {policy.get('type')!r} -- only 'MPC' is a policy here (evaluate.py:69)")
        if policy.get('collision_avoidance_type', 'circle') != 'circle':
            raise ValueError(f"{policy_yaml}: collision_avoidance_type {policy['collision_avoidance_type']!r} -- only the circle "
                             'constraint (mpc.py:223-226) is implemented')
        pt = str(policy.get('prediction_type', 'constant_acceleration'))
        if 'constant_acceleration' in pt:                                  # evaluate.py:76-81, in the reference's order
            kw['constant_speed'] = False
        elif 'constant_speed' in pt:
            kw['constant_speed'] = True
        else:
            raise ValueError('Invalid prediction type')
        if 'N' in policy:
            kw['N'] = int(policy['N'])
        if 'a_min' in policy:
            kw['a_min_policy'] = float(policy['a_min'])
    if env_yaml is not None:
        with open(env_yaml) as f:
            env = yaml.safe_load(f) or {}
        frozen = {'road_width': R.ROAD_WIDTH, 'road_length': R.ROAD_LENGTH, 'ca_radius': R.CA_RADIUS, 'num_agents': 2,
                  'v_max': 20.0, 'v_min': -2.0}
        for k, want in frozen.items():
            if k in env and abs(float(env[k]) - want) > 1e-12:
                raise ValueError(f'{env_yaml}: {k} = {env[k]} but the route tables / forecast of this package were generated for '
                                 f'{want} (igtmpc/data/route_constants.json, tests/golden/make_golden.py)')
        if 'dt' in env:
            kw['dt'] = float(env['dt'])
        if 'v0' in env:
            kw['v0'] = float(env['v0'])
        lim = {k: float(env[k]) for k in ('l_r', 'l_f') if k in env}
        if lim:
            kw['limits'] = lim
    if 'dt' in policy and 'dt' in kw and abs(float(policy['dt']) - kw['dt']) > 1e-12:
        raise ValueError(f"mpc.yaml dt = {policy['dt']} and fourwayint.yaml dt = {kw['dt']} disagree")
    return kw, policy


def save_results(r, save_dir, eval_mode, sc, seed=2026, policy=None, timenow=None):
    """The run directory the reference's driver leaves behind (evaluate.py:319-369 gt_mpc, 578-640 mpc), from the result of
    run_closed_loop -- so that what reads the reference's runs (generate_video.py:180-191 takes cl_traj.pkl[index] as
    [7 M, T+1] rows x, y, s, ey, epsi, v, heading per agent) reads these:

        <save_dir><eval_mode>_sc<sc>_seed<seed>_<time>/mpc/                  cl_traj.pkl [E, 7M, T+1], u_cl.pkl [E, 2M, T],
                                                                             eval_stats.csv, evaluation_data.pkl, mpc.yaml
        <save_dir><eval_mode>_sc<sc>_seed<seed>_<time>/game_mpc/evaluation/  cl_traj.pkl, u_cl.pkl, stats.csv, mpc.yaml

    `save_dir` is joined the way the reference joins it (plain concatenation: give it a trailing separator).  One csv row per
    episode with the reference's columns; the solve-time columns hold the lock-step batch's step time in seconds for every
    agent (the reference sums IPOPT's t_* statistics per agent, evaluate.py:294-297), its standard deviation over the steps
    (the reference's mpc branch takes the std of the mean, i.e. 0 -- evaluate.py:604).  Not written: the .mp4 (the directory is made) and the `refs`
    entry of evaluation_data.pkl (ReferenceGen's Euler-rolled path arrays, a set-up-time product outside this package).
    -> the run directory."""
    import csv
    import datetime
    import pickle

    import yaml
    gt = eval_mode == 'gt_mpc'
    timenow = timenow or datetime.datetime.now().strftime('%Y%m%d_%H%M%S')                     # evaluate.py:66
    run = save_dir + eval_mode + '_sc' + str(sc) + '_seed' + str(seed) + '_' + timenow
    out = run + ('/game_mpc/evaluation' if gt else '/mpc')
    os.makedirs(out, exist_ok=True)
    os.makedirs(run + ('/game_mpc/evaluation_videos' if gt else '/mpc/evaluation_videos'), exist_ok=True)
    x_cl, u_cl = np.asarray(r['x_data']), np.asarray(r['u_data'])
    E, M = x_cl.shape[0], x_cl.shape[1] // 7
    with open(out + '/mpc.yaml', 'w') as f:
        yaml.safe_dump(dict(policy or {}), f)
    # Runs append (evaluate.py:341-350).  The reference reads its own pickles back for that; this writer NEVER unpickles a
    # file (a run directory may come from anywhere, and unpickling executes what the file says): what it has written so far is
    # kept beside the pickles as plain arrays (.igt_run.npz, read with allow_pickle=False), the pickles are output only, and a
    # directory that holds pickles but not that file -- one this package did not write -- is refused.
    side = out + '/.igt_run.npz'
    new = {'x_cl': x_cl, 'u_cl': u_cl, 'weights': np.ones((E, 2)), 'routes': np.array([list(p) for p in r['routes']]),
           'deadlock': np.asarray(r['deadlock']).reshape(E, 1), 'initial_agents': x_cl[:, :, 0].reshape(E, M, 7)}
    if os.path.isfile(side):
        with np.load(side, allow_pickle=False) as old:
            acc = {k: np.concatenate([old[k], new[k]], axis=0) for k in new}
    elif any(os.path.isfile(out + '/' + n) for n in ('cl_traj.pkl', 'u_cl.pkl', 'evaluation_data.pkl')):
        raise FileExistsError(f'{out} holds pickles this package did not write (no .igt_run.npz beside them): it does not '
                              f'unpickle files to append to them; give another --save_dir')
    else:
        acc = new
    np.savez(side, **acc)
    for name, arr in (('cl_traj.pkl', acc['x_cl']), ('u_cl.pkl', acc['u_cl'])):
        with open(out + '/' + name, 'wb') as f:
            pickle.dump(arr, f)
    step_s = np.asarray(r['solve_ms'], dtype=np.float64) / 1e3
    rows = []
    for e in range(E):
        row = {'avg_sol_times': np.full(M, step_s.mean()), 'std_solve_times': [float(step_s.std())] * M,
               'infeasible_ratio': np.asarray(r['infeasible_ratio'][e]), 'deadlock': bool(r['deadlock'][e])}
        rows.append(dict({'NN_query_time': np.array([-1])}, **row) if gt else row)               # evaluate.py:355, 605
    csv_file = out + ('/stats.csv' if gt else '/eval_stats.csv')
    fresh = not os.path.isfile(csv_file)
    with open(csv_file, mode='w' if fresh else 'a', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        if fresh:
            w.writeheader()
        w.writerows(rows)
    if not gt:                                                                                 # evaluate.py:592-602
        data = dict({'N': (policy or {}).get('N'), 'agent_types': ['CAV'] * M}, **acc)
        with open(out + '/evaluation_data.pkl', 'wb') as f:
            pickle.dump(data, f, protocol=pickle.HIGHEST_PROTOCOL)
    return run


def main():
    ap = argparse.ArgumentParser(description='batched closed-loop evaluation (counterpart of evaluate.py --eval_mode mpc)')
    ap.add_argument('--sc', type=int, default=1)
    ap.add_argument('--num_samples', type=int, default=1)
    ap.add_argument('--N', type=int, default=None, help='horizon; default: the policy file\'s, else 40 (mpc.yaml:6; BASELINE.json benchmarks 20)')
    ap.add_argument('--policy_config', default=None, help="the reference's mpc.yaml (evaluate.py:32)")
    ap.add_argument('--env_config', default=None, help="the reference's common/fourwayint.yaml (evaluate.py:28)")
    ap.add_argument('--C', type=int, default=256)
    ap.add_argument('--eval_mode', default='mpc', choices=['mpc', 'gt_mpc'])
    ap.add_argument('--value_net', default=None, help='gt_mpc: .npz with W0,b0,W1,b1,... (optionally prefixed, see --net_prefix); '
                    'default: the network shipped for --sc')
    ap.add_argument('--net_prefix', default='', help="key prefix inside the npz, e.g. 'sc1_'")
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--device_resident', action='store_true', help='keep all per-step arrays in HBM (torch tensors)')
    ap.add_argument('--graph', action='store_true', help='with --device_resident: capture one step in a stream graph and replay it')
    ap.add_argument('--cand_mode', default='track', choices=['lattice', 'ramp_hold', 'track'])
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--save_dir', default=None, help='write the run directory the reference\'s driver writes (cl_traj.pkl, u_cl.pkl, '
                    'stats csv, mpc.yaml; evaluate.py:646) under <save_dir><eval_mode>_sc<sc>_seed2026_<time>/')
    a = ap.parse_args()
    net = None
    if a.eval_mode == 'gt_mpc':
        if a.value_net:
            z = np.load(a.value_net)
            layers, i = [], 0
            while f'{a.net_prefix}W{i}' in z:
                layers.append((z[f'{a.net_prefix}W{i}'], z[f'{a.net_prefix}b{i}']))
                i += 1
            net = dict(layers=layers)
    kw, policy_file = load_reference_configs(a.policy_config, a.env_config)
    if a.N is not None:
        kw['N'] = a.N
    kw.setdefault('N', 40)
    a.N = kw['N']
    r = run_closed_loop(sc=a.sc, num_samples=a.num_samples, C=a.C, verbose=a.verbose, eval_mode=a.eval_mode,
                        value_net=net, device_resident=a.device_resident, cand_mode=a.cand_mode, dtype=a.dtype, graph=a.graph, **kw)
    if a.save_dir is not None:
        policy = {'type': 'MPC', 'N': a.N, 'dt': kw.get('dt', 0.1), 'a_min': -4, 'a_max': 3, 'v_min': -1.0, 'v_max': 5,      # mpc.yaml's keys
                  'prediction_type': 'constant_acceleration', 'collision_avoidance_type': 'circle', **policy_file, 'N': a.N,
                  'solver': {'library': 'igtmpc', 'cand_mode': a.cand_mode, 'C': a.C, 'dtype': a.dtype, 'track_env': r.get('track_env')}}
        r['run_dir'] = save_results(r, a.save_dir, a.eval_mode, a.sc, policy=policy)
    print(json.dumps({'sc': a.sc, 'episodes': a.num_samples, 'routes': r['routes'][:4],
                      'infeasible_ratio_mean': r['infeasible_ratio'].mean(axis=0).tolist(),
                      'deadlock_rate': float(r['deadlock'].mean()),
                      'final_s_mean': r['x_data'][:, 2::7, -1].mean(axis=0).tolist(),
                      'avg_solve_ms_per_step': float(r['solve_ms'][5:].mean()), 'run_dir': r.get('run_dir')}))


if __name__ == '__main__':
    main()
