"""MPC_Planner -- same constructor keywords, methods and return contract as the reference
planner (mpc.py:19-406), so an evaluate.py-style driver can swap it in.

What changes underneath: no CasADi NLP is built and no IPOPT runs.  `solve` hands the current
initial condition and predictions to the batched shooting solver in libigtmpc.so (one wavefront
per (scenario, ego) problem on the MI355X) and returns the feasible arg-min of the candidate set.
Planners created with the same discretisation share one solver handle per device; the batched
entry (`BatchSolver.solve`) is the throughput path, this class is the per-agent compatibility face.
"""
import atexit
import hashlib
import os
import re
import time
import warnings

import numpy as np

from .cinf import cinf_halfplanes
from ._lib import IGT_FLAG_WARM
from .solver import BatchSolver
from .value_nets import shipped_value_net
from .vehicle import Curvature, VehicleAction, VehicleReference  # noqa: F401  (re-exported for drivers)

_SHARED = {}


@atexit.register
def close_shared_solvers():
    """Planners share solver handles (one per configuration); they are destroyed here, once, at interpreter exit --
    or earlier by a driver that is done with its planners."""
    while _SHARED:
        _SHARED.popitem()[1].close()


def value_net_from_config(nn_config_dir):
    """What mpc.py:72-74, 108-124 does with `nn_config_dir` (evaluate.py:191 passes get_scenario_config(sc) =
    <cwd>/game_theoretic_NN/configs/sc{n}_config.yaml): read the YAML, build mlp(6, 1, [hidden_size]*num_layers, tanh)
    and load `model_path`.  Here: the scenario number is taken from the YAML's model_path (V_GT_sc{n}.pt) or, when the
    file is not there, from the path's own name (sc{n}_config); a checkpoint found at <cwd> + model_path is read with
    torch.load(weights_only=True), otherwise the weights shipped with this package are used (the same eight networks).
    The feature / target statistics (mpc.py:110-118) live in a dataset pickle the reference does not ship: identity
    whitening is used and a warning says so.  -> (value_net dict for BatchSolver.set_value_net, include_route)."""
    cfg = None
    if os.path.isfile(nn_config_dir):
        import yaml
        with open(nn_config_dir) as f:
            cfg = yaml.safe_load(f) or {}
    m = re.search(r'V_GT_sc(\d+)', str((cfg or {}).get('model_path', ''))) or \
        re.search(r'sc(\d+)_config', os.path.basename(str(nn_config_dir)))
    if m is None:
        raise ValueError(f'nn_config_dir={nn_config_dir!r}: neither a readable sc<n>_config.yaml nor a path that names one')
    sc = int(m.group(1))
    layers = None
    ckpt = os.getcwd() + str((cfg or {}).get('model_path', ''))             # mpc.py:124
    if cfg and os.path.isfile(ckpt):
        import torch
        sd = torch.load(ckpt, map_location='cpu', weights_only=True)
        keys = sorted((k for k in sd if k.endswith('weight')), key=lambda k: int(k.split('.')[1]))
        layers = [(sd[k].double().numpy(), sd[k.replace('weight', 'bias')].double().numpy()) for k in keys]
    net = dict(layers=layers) if layers is not None else shipped_value_net(sc)
    if cfg:
        n_hidden, width = len(net['layers']) - 1, net['layers'][0][0].shape[0]
        if int(cfg.get('num_layers', n_hidden)) != n_hidden or int(cfg.get('hidden_size', width)) != width:
            raise ValueError(f'{nn_config_dir}: hidden_size / num_layers do not describe the scenario-{sc} network '
                             f'({n_hidden} hidden layers of {width})')
    warnings.warn('gt_mpc value network: the feature / target normalisation statistics (mpc.py:110-118) come from a '
                  'dataset the reference does not ship; identity statistics are used (pass value_net=dict(layers, Wn, '
                  'mu_f, sigma_t, mu_t) to supply them)', stacklevel=3)
    return net, bool((cfg or {}).get('include_route', False))


def _net_digest(value_net):
    if value_net is None:
        return None
    h = hashlib.sha256()
    for W, b in value_net['layers']:
        h.update(np.ascontiguousarray(W, dtype=np.float64).tobytes())
        h.update(np.ascontiguousarray(b, dtype=np.float64).tobytes())
    for k in ('Wn', 'mu_f', 'sigma_t', 'mu_t'):
        if value_net.get(k) is not None:
            h.update(np.ascontiguousarray(value_net[k], dtype=np.float64).tobytes())
    return h.hexdigest()


def augment_prev_sol(data, model, K):
    """utils.py:354-363: previous solution (x[7,N+1], u[2,N]) -> warm start of the next solve: shifted by one step, the
    states extended by one model step with the last control (retried with a = 0 when that step ends above v = 5,
    v clipped to [-1, 5]), the controls by repeating the last one."""
    x_sol_prev, u_sol_prev = np.asarray(data[0], dtype=np.float64), np.asarray(data[1], dtype=np.float64)
    last = dict(zip(('x', 'y', 's', 'ey', 'epsi', 'v', 'heading'), x_sol_prev[:, -1]), K=K)
    nxt = model(VehicleReference(last), VehicleAction({'a': u_sol_prev[0, -1], 'df': u_sol_prev[1, -1]}))
    if nxt.v > 5:
        nxt = model(VehicleReference(last), VehicleAction({'a': 0, 'df': u_sol_prev[1, -1]}))
    col = np.array([[nxt.x], [nxt.y], [nxt.s], [nxt.ey], [nxt.epsi], [np.clip(nxt.v, -1, 5)], [nxt.heading]])
    return (np.hstack([x_sol_prev[:, 1:], col]), np.hstack([u_sol_prev[:, 1:], u_sol_prev[:, [-1]]]))


class _Stats:
    """Stand-in for casadi's OptiSol as far as evaluate.py reads it (evaluate.py:294-295, 549-552)."""

    def __init__(self, t, status, cost, argmin):
        self._d = {'t_wall_total': t, 'success': status == 0, 'return_status': 'Solve_Succeeded' if status == 0 else
                   'Infeasible_Problem_Detected', 'cost': cost, 'argmin': argmin}

    def stats(self):
        return dict(self._d)


class MPC_Planner:
    def __init__(self, N=10, dt=0.1, agents=None, goals=None, ca_radius=2.8, ref=None, road_dim=(10, 50),
                 routes=None, ds_right=None, index=None, num_rk4_steps=7, solver='ipopt', ca_type='circle',
                 nn_config_dir=None, use_NN_cost2go=False, weights=(1, 1, 1),
                 C=256, device=0, dtype='f64', value_net=None, cand_mode='track', refine_iters=0, track_env=None,
                 warm_start=None):
        assert agents is not None, 'Agents are not defined'           # mpc.py:155
        assert index is not None                                      # mpc.py:80
        if ca_type != 'circle':
            # OBCA needs dual variables as decision variables (mpc.py:211-221): not a shooting constraint
            raise NotImplementedError("only ca_type='circle' (mpc.yaml:16) is supported")
        self.N, self.dt, self.ca_type = N, dt, ca_type
        self.x_sol_prev = None
        self.d_min = 2 * ca_radius                                    # mpc.py:45
        self.l = 4.47; self.l_f = 4.47 / 2; self.l_r = 4.47 / 2; self.width = 2.0      # mpc.py:48-51
        self.routes = routes
        self.steering_rate_limit = 0.7; self.jerk_limit = 0.9         # mpc.py:55-56
        self.v_min, self.v_max, self.a_min, self.a_max = 0, 5, -4, 3  # mpc.py:57-60
        self.ey_lim, self.max_steering = 0.2, 1                       # mpc.py:61-62
        self.use_NN_cost2go = use_NN_cost2go
        self.road_width, self.road_length = road_dim                  # mpc.py:65-66
        self.ref, self.ds_right, self.weights, self.goals = ref, ds_right, weights, goals
        self.nx, self.nu = 7, 2
        self.agents, self.ind = agents, index
        self.initial_agent = agents[index]
        self.M = len(ref) if ref is not None else len(agents)         # mpc.py:82
        self.num_obstacles = self.M - 1
        self.pred_ind = [i for i in range(self.M) if i != index]
        self.NN_query_time = -1
        self.solve_time = 0.0
        self.sol = None
        # curvature of the ego route (mpc.py:183-200)
        route = routes[index]
        if ref is not None and 'K' in ref[index]:
            self.K = Curvature.from_reference(ref[index]['K'], route, road_dim, ds_right if ds_right is not None else
                                              self.road_width - ca_radius)
        else:
            self.K = Curvature.from_route(route)
        self._abs_heading = [r in ('32', '41') for r in routes]       # mpc.py:231, 250, 273, 282
        # terminal set (mpc.py:88-104) -- same for every planner with this dt
        self.C_inf = cinf_halfplanes(dt=dt, jerk=self.jerk_limit)
        cost_mode = 'value_net' if use_NN_cost2go else 'progress'
        self.cand_mode = cand_mode
        # whether solve() centres the candidates on u_sol_prev: default on for ramp-hold, off for the tracking family
        # (igtmpc.evaluate.run_closed_loop has the closed-loop figures behind that default)
        self.warm_start = (cand_mode == 'ramp_hold') if warm_start is None else (bool(warm_start) and cand_mode in ('ramp_hold', 'track'))
        include_route = False
        if use_NN_cost2go and value_net is None:
            if nn_config_dir is None:
                raise ValueError('use_NN_cost2go=True needs nn_config_dir (the reference\'s call, evaluate.py:191) or '
                                 'value_net=dict(layers[, Wn, mu_f, sigma_t, mu_t])')
            value_net, include_route = value_net_from_config(nn_config_dir)     # mpc.py:72-74, 108-124
        self.NN_config = dict(include_route=include_route) if use_NN_cost2go else None
        # planners with the same discretisation, candidate family and value network share one solver handle; the
        # network is keyed by CONTENT (a dict rebuilt with the same weights maps to the same handle)
        if track_env is None:      # a planner lives in a closed loop: the envelope's scale follows the horizon (evaluate.py)
            from .evaluate import auto_track_env
            track_env = auto_track_env(N, dt)
        self.track_env = float(track_env)
        key = (N, dt, num_rk4_steps, C, self.num_obstacles, device, dtype, cost_mode, self.d_min, cand_mode, refine_iters,
               self.track_env, _net_digest(value_net))
        if key not in _SHARED:
            s = BatchSolver(N=N, dt=dt, n_rk4=num_rk4_steps, C=C, n_obs=self.num_obstacles, device=device,
                            dtype=dtype, cost_mode=cost_mode, d_min=self.d_min, cand_mode=cand_mode,
                            refine_iters=refine_iters, track_env=self.track_env)
            s.set_cinf(*self.C_inf)
            if use_NN_cost2go:
                s.set_value_net(**value_net)
            _SHARED[key] = s
        self._solver = _SHARED[key]
        self._dt = self._solver.np_dtype
        self._x0 = np.zeros((1, 7), self._dt)
        self._u_prev = np.zeros((1, 2), self._dt)
        self._obs = np.zeros((1, self.num_obstacles, 2, N + 1), self._dt)
        self._tv_sv = np.zeros((1, 2), self._dt)
        self._enc = np.zeros((1, 2), self._dt)
        if use_NN_cost2go:
            from .routes import ROUTE_ID, scenario_encoding_sign, scenario_of
            if include_route:
                e = [ROUTE_ID[r] for r in routes]                                    # mpc.py:332-334, utils.py:393-402
            else:
                e = scenario_encoding_sign(list(routes), scenario_of(list(routes)))  # mpc.py:336-337
            j = self.pred_ind[0]
            self._enc[0] = (e[index], e[j])
        self.update_initial_condition(self.initial_agent, VehicleAction({'a': 0.0, 'df': 0.0}))

    # ------------------------------------------------------------------ mpc.py:280-294
    def update_initial_condition(self, agent, u_prev):
        self.u_prev_raw = u_prev
        st = agent['state']
        self._x0[0] = (st.x, st.y, st.s, st.ey, st.epsi, st.v, st.heading)
        self._u_prev[0] = (u_prev.a, u_prev.df)
        self.initial_agent = agent

    # ------------------------------------------------------------------ mpc.py:241-278
    def update_predictions(self, preds, raw_preds=None):
        assert len(preds) == self.M, ValueError('Invalid number of predictions')
        m = 0
        self.pred_ind = []
        for i, pred in enumerate(preds):
            if i != self.ind:
                self.pred_ind.append(i)
                assert len(pred) == self.N + 1, ValueError('Invalid prediction length (Horizon)')
                # only x,y of the obstacle block are read by the cost / constraints (mpc.py:223-226)
                self._obs[0, m, 0] = [p.x for p in pred]
                self._obs[0, m, 1] = [p.y for p in pred]
                m += 1
        if raw_preds is not None:
            self.raw_preds_np = np.zeros((1, self.nx * self.M))
            for i, pred in enumerate(raw_preds):
                p = pred[-1]
                self.raw_preds_np[0, self.nx * i:self.nx * (i + 1)] = [p.x, p.y, p.s, p.ey, p.epsi, p.v, p.heading]
            j = self.pred_ind[0]                                       # mpc.py:329-330
            self._tv_sv[0] = (raw_preds[j][-1].s, raw_preds[j][-1].v)
        self.NN_query_time = -1                                        # mpc.py:278

    # ------------------------------------------------------------------ mpc.py:383-406
    def solve(self, x_sol_prev=None, u_sol_prev=None):
        """-> (x[7,N+1], u[2,N], True) or (None, None, False); never raises on an infeasible problem.
        u_sol_prev[2,N] -- what evaluate.py:478-482 passes after augment_prev_sol -- is the warm start: IPOPT started
        its iterations there (mpc.py:386-389), the shooting solver centres its candidates there when `warm_start` is on
        (default: ramp-hold yes, tracking no), so that the shifted previous plan is itself one of the candidates.
        x_sol_prev has no counterpart (states are implied)."""
        flags = np.array([1 if self._abs_heading[self.ind] else 0], dtype=np.uint32)
        kp = np.array([self.K.kparams], dtype=self._dt)
        u_ws = None
        if u_sol_prev is not None and self.warm_start:
            u_ws = np.ascontiguousarray(np.asarray(u_sol_prev, dtype=self._dt).reshape(1, 2, self.N))
            flags = flags | np.uint32(IGT_FLAG_WARM)
        t0 = time.time()
        out = self._solver.solve(self._x0, self._u_prev, kp, flags, self._obs,
                                 self._tv_sv if self.use_NN_cost2go else None,
                                 self._enc if self.use_NN_cost2go else None, u_ws=u_ws)
        self.solve_time = time.time() - t0
        status = int(out['status'][0])
        self.sol = _Stats(self.solve_time, status, float(out['cost'][0]), int(out['argmin'][0]))
        if status != 0:
            return (None, None, False)
        x = np.array(out['x'][0], dtype=np.float64)
        u = np.array(out['u'][0], dtype=np.float64)
        self.x_sol_prev = x
        return (x, u, True)

    def CAV_utility(self, x, u):
        """mpc.py:356-373 on a trajectory (x[7,N+1], u[2,N]) in float64 on the host -- the expression the kernels
        evaluate per candidate (progress cost; the value-net term lives on the device only)."""
        x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
        J = 0.0
        for k in range(self.N + 1):      # squares as products: what casadi's sq() and the kernels evaluate (a numpy
            if k < self.N:               # SCALAR ** 2 goes through libm pow(), one ulp off for ~1 value in 10^4)
                J = J + 0.05 * (u[0, k] * u[0, k] + u[1, k] * u[1, k])  # mpc.py:362
            J = J + x[4, k] * x[4, k]                                   # mpc.py:363
            J = J + x[3, k] * x[3, k]                                   # mpc.py:364
        return J - (x[2, self.N] - x[2, 0])                             # mpc.py:372

    def cost_function(self):
        """mpc.py:375-376 returns the CasADi expression of the cost; here: the cost of the last solution."""
        if self.sol is None:
            raise RuntimeError('cost_function() needs a solve first')
        return self.sol.stats()['cost']
