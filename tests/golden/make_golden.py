#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory by RUNNING THE REFERENCE's own
Python in the build container (it reads /root/reference, which never travels).
Only arrays / JSON numbers are written; no reference source is copied.

    python tests/golden/make_golden.py          # rewrites the fixtures

What runs: the numpy branches of
  common/kinematic_bicycle_model_frenet.py:70-127   (RK4 Frenet model)
  common/kinematic_bicycle_model.py:15-50            (Cartesian Euler model)
  common/ReferenceGen.py:41-235                      (route geometry -> K, radii, headings)
  model.py:14-51 + game_theoretic_NN/models/*.pt     (terminal value MLP, torch CPU fp64)
Those files begin with `import casadi as ca` but their numpy branches never touch
`ca`; casadi is not installed here, so the import line is satisfied with an EMPTY
placeholder module object (no attributes, nothing emulated).

mpc.py and common/utils.py cannot be IMPORTED (`import polytope`; utils.py:599 evaluates
`pt.Polytope()` at import).  Their functions that never call casadi / polytope are run
one by one instead (round 3): the FunctionDef node is taken out of the file's syntax
tree and compiled UNMODIFIED, under its own file name and line numbers, into a namespace
that holds only what the body names (numpy, copy, typing.List, the reference's own
VehicleReference / VehicleAction classes).  `self` is a SimpleNamespace carrying numpy
`x`, `u`; `opti.subject_to` / `opti.set_value` are list sinks that record what they are
handed -- nothing of casadi is emulated.  Run that way:
  mpc.py:356-373   MPC_Planner.CAV_utility (progress cost)            -> cost_golden.npz
  mpc.py:296-321   add_ey / add_input_rate / add_state_and_input_constraints
  mpc.py:177-180   add_terminal_constraints (index convention)        -> verdict_golden.npz
  mpc.py:241-294   update_predictions / update_initial_condition      -> marshalling_golden.npz
  mpc.py:326-354   get_xN(mode='numpy')                               -> value_features_golden.npz
  utils.py:84-169  scenario_index / scenario_encoding; 393-402 route_encoding -> scenario_encoding.json
  utils.py:365-388 filter_preds                                       -> filter_preds_golden.npz
  utils.py:354-363 augment_prev_sol (with the reference's numpy model)-> augment_prev_sol_golden.npz
  utils.py:532-586 frenet2global, straight routes (no casadi call)    -> frenet2global_straight_golden.npz
What would actually call casadi / polytope stays "parity unpinned" (DESIGN.md section 5):
the collision constraint (ca.bilin), pw_const at s == b, the C_inf table (polytope),
frenet2global on turning routes / the heading ramp (ca.if_else, ca.pw_lin), the predictor
and share_motion_forecasts (ca.SX.sym), IPOPT's optimum.
"""
import ast
import copy
import json
import os
import sys
import types
from typing import List

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit('reference not mounted; fixtures can only be regenerated in the build container')
    sys.modules.setdefault('casadi', types.ModuleType('casadi'))   # empty, never called
    sys.path.insert(0, os.path.join(REF, 'common'))
    sys.path.insert(0, REF)
    from kinematic_bicycle_model_frenet import KinematicBicycleModelFrenet
    from kinematic_bicycle_model import KinematicBicycleModel
    from VehicleReference import VehicleReference
    from VehicleAction import VehicleAction
    from VehicleState import VehicleState
    from ReferenceGen import ReferenceGenerator
    return dict(Frenet=KinematicBicycleModelFrenet, Cart=KinematicBicycleModel, Ref=VehicleReference,
                Act=VehicleAction, State=VehicleState, RefGen=ReferenceGenerator)


def make_K(b0, b1, kv):
    """plain-Python stand-in for ca.pw_const(s,[b0,b1],[0,kv,0]) (mpc.py:199)."""
    def K(s):
        return (kv if s >= b0 else 0.0) - (kv if s >= b1 else 0.0)
    return K


LEFT = (19.3, 19.3 + 8.6 * np.pi / 2, 1 / 8.6)
RIGHT = (10.7, 10.7 + 11.4 * np.pi / 2, -1 / 11.4)
STRAIGHT = (np.inf, np.inf, 0.0)


def frenet_cases(rng, n):
    """(x0[n,7] planner order, U[n,2,20], kp[n,3]) covering break-point straddlers,
    negative curvature, |ey| near the 0.2 limit, v near 0 and 5."""
    N = 20
    x0 = np.zeros((n, 7))
    U = np.zeros((n, 2, N))
    kp = np.zeros((n, 3))
    for i in range(n):
        kind = i % 8
        route = (LEFT, RIGHT, STRAIGHT)[i % 3]
        s0 = rng.uniform(0, 45)
        v0 = rng.uniform(0.5, 4.5)
        ey0 = rng.uniform(-0.1, 0.1)
        ep0 = rng.uniform(-0.05, 0.05)
        if kind == 1 and route is not STRAIGHT:        # enters the arc inside the horizon
            s0 = route[0] - rng.uniform(0.0, 3.0)
        elif kind == 2 and route is not STRAIGHT:      # leaves the arc inside the horizon
            s0 = route[1] - rng.uniform(0.0, 3.0)
        elif kind == 3 and route is not STRAIGHT:      # starts within a stage-length of a break-point
            s0 = route[i % 2] - rng.uniform(0.0, 0.08)
        elif kind == 4:
            ey0 = rng.choice([-1, 1]) * rng.uniform(0.18, 0.22)
        elif kind == 5:
            v0 = rng.uniform(0.0, 0.2)
        elif kind == 6:
            v0 = rng.uniform(4.8, 5.2)
        x0[i] = [rng.uniform(0, 50), rng.uniform(-20, 30), s0, ey0, ep0, v0, rng.uniform(-np.pi, np.pi)]
        a = rng.uniform(-1, 1)
        d = rng.uniform(-0.1, 0.1)
        da = rng.uniform(-0.09, 0.09)
        dd = rng.uniform(-0.07, 0.07)
        for k in range(N):
            if kind == 7:                               # rough, non-lattice controls
                a = np.clip(a + rng.uniform(-0.09, 0.09), -4, 3)
                d = np.clip(d + rng.uniform(-0.07, 0.07), -1, 1)
            else:
                a = np.clip(a + da, -4, 3)
                d = np.clip(d + dd, -1, 1)
            U[i, 0, k], U[i, 1, k] = a, d
        kp[i] = route
    return x0, U, kp


def run_frenet(R, x0, U, kp, n_rk4=4, dt=0.1):
    model = R['Frenet'](2.235, 2.235, 2.0, dt, discretization='rk4', mode='numpy', num_rk4_steps=n_rk4)
    n, _, N = U.shape
    X = np.zeros((n, 7, N + 1))
    for i in range(n):
        K = make_K(*kp[i])
        st = R['Ref']({'x': x0[i, 0], 'y': x0[i, 1], 's': x0[i, 2], 'ey': x0[i, 3], 'epsi': x0[i, 4],
                       'v': x0[i, 5], 'heading': x0[i, 6], 'K': K})
        X[i, :, 0] = x0[i]
        for k in range(N):
            st = model(st, R['Act']({'a': U[i, 0, k], 'df': U[i, 1, k]}))
            X[i, :, k + 1] = [st.x, st.y, st.s, st.ey, st.epsi, st.v, st.heading]
    return X


def run_cartesian(R, rng, n):
    model = R['Cart'](2.235, 2.235, 2.0, 0.1)
    z = np.column_stack([rng.uniform(0, 50, n), rng.uniform(-20, 30, n), rng.uniform(-np.pi, np.pi, n),
                         rng.uniform(0, 6, n)])
    u = np.column_stack([rng.uniform(-4, 3, n), rng.uniform(-1, 1, n)])
    out = np.zeros_like(z)
    for i in range(n):
        st = R['State']({'x': z[i, 0], 'y': z[i, 1], 'heading': z[i, 2], 'v': z[i, 3]})
        nx = model(st, R['Act']({'a': u[i, 0], 'df': u[i, 1]}))
        out[i] = [nx.x, nx.y, nx.heading, nx.v]
    return z, u, out


def route_constants(R):
    """Runs ReferenceGenerator exactly as evaluate.py:38-119 does, one route at a time,
    and records the numbers the hot path consumes."""
    road_width, road_length, ca_radius, dt, v_des = 11.4, 50, 2.8, 0.1, 5       # fourwayint.yaml
    fillet = road_width - ca_radius                                             # evaluate.py:48
    VS = R['State']
    goals_states = {'1': VS({'x': 0, 'y': road_width - ca_radius, 'heading': -np.pi, 'v': v_des}),
                    '2': VS({'x': road_length / 2 + road_width / 2 - ca_radius, 'y': road_length / 2 + road_width / 2, 'heading': np.pi / 2, 'v': v_des}),
                    '3': VS({'x': road_length, 'y': ca_radius, 'heading': 0, 'v': v_des}),
                    '4': VS({'x': road_length / 2 - road_width / 2 + ca_radius, 'y': (road_width - road_length) / 2, 'heading': -np.pi / 2, 'v': v_des})}
    init_states = {'1': VS({'x': 0, 'y': ca_radius, 'heading': 0, 'v': 0}),
                   '2': VS({'x': road_length / 2 - road_width / 2 + ca_radius, 'y': road_length / 2 + road_width / 2, 'heading': -np.pi / 2, 'v': 0}),
                   '3': VS({'x': road_length, 'y': road_width - ca_radius, 'heading': -np.pi, 'v': 0}),
                   '4': VS({'x': road_length / 2 + road_width / 2 - ca_radius, 'y': (road_width - road_length) / 2, 'heading': np.pi / 2, 'v': 0})}
    out = {}
    M_sim = 150
    for o in '1234':
        for g in '1234':
            if o == g:
                continue
            route = o + g
            init = [{'type': 'CAV', 'state': init_states[o]}]
            gen = R['RefGen'](N=2 * M_sim, dt=dt, initial_state=init, goals=[goals_states[g]], env=None,
                              radius=ca_radius, routes=[route], target_velocity=v_des, mode='frenet',
                              road_width=road_width, road_length=road_length, fillet_radius=fillet)
            ref = gen.get_reference(M_sim, initial_states=init, output_type=dict)[0]
            K = ref['K']
            nz = np.nonzero(K)[0]
            rec = dict(x0=float(ref['x'][0]), y0=float(ref['y'][0]), xN=float(ref['x'][-1]), yN=float(ref['y'][-1]),
                       heading0=float(ref['heading'][0]), headingN=float(ref['heading'][-1]),
                       straight=bool(len(nz) == 0))
            if len(nz):
                radius = float(max(abs(1 / K[nz])))                       # mpc.py:190
                kv = float(K[nz[0]])                                      # mpc.py:193
                if route in ['12', '23', '34', '41']:
                    b0 = (road_length - road_width) / 2                   # mpc.py:190
                else:
                    b0 = (road_length - road_width) / 2 - fillet          # mpc.py:192
                rec.update(Kv=kv, radius=radius, b0=float(b0), b1=float(b0 + radius * np.pi / 2))
            out[route] = rec
    return out


def value_nets(rng):
    import torch
    from model import mlp                                                   # reference model.py:14-51
    z = rng.normal(size=(64, 6))
    out = {'z': z}
    for sc in range(1, 9):
        sd = torch.load(f'{REF}/game_theoretic_NN/models/V_GT_sc{sc}.pt', map_location='cpu', weights_only=True)
        n_lin = len([k for k in sd if k.endswith('weight')])
        net = mlp(input_layer_size=6, output_layer_size=1, hidden_layer_sizes=[128] * (n_lin - 1),
                  activation='tanh', batch_norm=False)
        net.load_state_dict(sd)
        net.eval()
        with torch.no_grad():
            out[f'V_sc{sc}'] = net(torch.tensor(z)).numpy()[:, 0]
        if sc in (1, 3):       # one 2-hidden-layer and one 3-hidden-layer net travel as data
            keys = sorted([k for k in sd if k.endswith('weight')], key=lambda s: int(s.split('.')[1]))
            for li, k in enumerate(keys):
                out[f'sc{sc}_W{li}'] = sd[k].numpy()
                out[f'sc{sc}_b{li}'] = sd[k.replace('weight', 'bias')].numpy()
    return out


# ---------------------------------------------------------------------------------------------
# round 3: functions of mpc.py / common/utils.py executed one by one (see the module docstring)
# ---------------------------------------------------------------------------------------------
def extract_functions(path, names, cls=None, namespace=None):
    """The named FunctionDef nodes of the reference file `path` (module level, or methods of class `cls`), compiled
    unmodified -- same file name, same line numbers, so a traceback cites the reference -- and executed into
    `namespace`.  -> dict name -> function."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    picked = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise SystemExit(f'{path}: no such function(s) {sorted(missing)}')
    ns = dict(namespace or {})
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, 'exec'), ns)
    return {n: ns[n] for n in names}


class Sink:
    """What stands where the planner's casadi `Opti` object would: it keeps what it is handed."""

    def __init__(self):
        self.constraints = []       # subject_to(expr): with numpy x, u the expression is already a bool
        self.values = {}            # set_value(param, value)

    def subject_to(self, expr):
        self.constraints.append(bool(expr))

    def set_value(self, param, value):
        self.values[param] = value


class ParamKeys:
    """Parameter handle of the sink: indexing returns a hashable key, `self.x0[6]` -> ('x0', 6)."""

    def __init__(self, name):
        self.name = name

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            idx = tuple('all' if isinstance(i, slice) else int(i) for i in idx)
        return (self.name, idx)

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return isinstance(other, ParamKeys) and other.name == self.name


def reference_functions(R):
    utils = extract_functions(f'{REF}/common/utils.py',
                              ['scenario_index', 'scenario_encoding', 'route_encoding', 'filter_preds', 'augment_prev_sol',
                               'frenet2global'],
                              namespace=dict(np=np, copy=copy, List=List, VehicleReference=R['Ref'],
                                             VehicleAction=R['Act']))
    mpc = extract_functions(f'{REF}/mpc.py',
                            ['CAV_utility', 'add_ey_constraints', 'add_input_rate_constraints',
                             'add_state_and_input_constraints', 'add_terminal_constraints', 'update_predictions',
                             'update_initial_condition', 'get_xN'],
                            cls='MPC_Planner',
                            namespace=dict(np=np, scenario_encoding=utils['scenario_encoding'],
                                           route_encoding=utils['route_encoding']))
    return utils, mpc


def planner_self(N=20, dt=0.1, **kw):
    """The attributes MPC_Planner.__init__ sets before it builds the NLP (mpc.py:38-62), as plain numbers."""
    d = dict(N=N, dt=dt, steering_rate_limit=0.7, jerk_limit=0.9, v_min=0, v_max=5, a_min=-4, a_max=3, ey_lim=0.2,
             max_steering=1, use_NN_cost2go=False, nx=7, nu=2, opti=Sink())
    d.update(kw)
    return types.SimpleNamespace(**d)


def reference_rollouts():
    """The committed golden rollouts (made above by the reference's model): real trajectories to cost and judge."""
    g = np.load(os.path.join(HERE, 'frenet_rk4_golden.npz'))
    return g['x0'], g['U'], g['kp'], np.transpose(g['X'], (0, 1, 2))


def cost_fixture(mpc, rng):
    """mpc.py:356-373 on (a) the 1024 golden rollouts, (b) 256 random arrays at N = 20, (c) 32 at N = 40 and N = 10."""
    out = {}
    x0, U, kp, X = reference_rollouts()
    sets = {'rolled': (X, U)}
    for N, n in ((20, 256), (40, 32), (10, 32)):
        sets[f'random_N{N}'] = (rng.normal(size=(n, 7, N + 1)) * np.array([20, 20, 20, 0.3, 0.3, 3, 2])[None, :, None],
                                rng.uniform(-1, 1, size=(n, 2, N)) * np.array([4, 1])[None, :, None])
    for name, (Xs, Us) in sets.items():
        N = Us.shape[-1]
        # the body runs once per set on arrays with the batch as TRAILING axis (self.x[4,k] is then a vector):
        # element by element, `np.float64 ** 2` goes through libm pow(), which is one ulp away from x*x for about one
        # value in 10^4 (seen here: 1 of 256 costs differed by 1.8e-15), while casadi's `x**2` is sq(x) = x*x -- the
        # array form is the arithmetic the reference's NLP actually evaluates
        J = mpc['CAV_utility'](planner_self(N=N, x=np.moveaxis(Xs, 0, -1), u=np.moveaxis(Us, 0, -1)))
        if name != 'rolled':
            out[f'{name}_X'], out[f'{name}_U'] = Xs, Us
        out[f'{name}_J'] = J
    return out


def verdict_bits(mpc, X, U, u_prev, A, b, N, dt=0.1):
    """The reference's constraint builders on one numpy trajectory -> bit mask of the VIOLATED families, numbered like
    oracle/np_oracle.constraint_violation: 0 box v (mpc.py:316-317), 1 box a, df (318-321), 2 rate (301-312),
    3 |ey| (296-299), 4 terminal set (177-180)."""
    bits = 0
    me = planner_self(N=N, dt=dt, x=X, u=U, u_prev=u_prev, C_inf=types.SimpleNamespace(A=A, b=b))
    me.opti = Sink()
    mpc['add_state_and_input_constraints'](me)
    c = np.array(me.opti.constraints).reshape(N, 6)                 # per k: v>=, v<=, a>=, a<=, df>=, df<=
    bits |= (0 if c[:, :2].all() else 1) | (0 if c[:, 2:].all() else 2)
    me.opti = Sink()
    mpc['add_input_rate_constraints'](me)
    bits |= 0 if all(me.opti.constraints) else 4
    assert len(me.opti.constraints) == 4 * N
    me.opti = Sink()
    mpc['add_ey_constraints'](me)
    bits |= 0 if all(me.opti.constraints) else 8
    assert len(me.opti.constraints) == 2 * (N + 1)
    me.opti = Sink()
    mpc['add_terminal_constraints'](me)
    bits |= 0 if all(me.opti.constraints) else 16
    assert len(me.opti.constraints) == len(b)
    return bits


def verdict_fixture(mpc, rng):
    """(a) `rolled`: the golden rollouts with a drawn u_prev; (b) `edge`: hand-made trajectories that sit ON each
    threshold and each k-range edge (v_N > 5 is accepted: box is k < N; |ey_N| > 0.2 is rejected: k <= N; the terminal
    set reads (v_{N-1}, a_{N-1}); a rate step of exactly dt*jerk is accepted)."""
    sys.path.insert(0, os.path.join(HERE, '..', '..', 'igt-mpc-int_amd'))
    from igtmpc.cinf import cinf_halfplanes               # pure numpy; the C_inf TABLE is the product's own (unpinned)
    A, b = cinf_halfplanes(dt=0.1, jerk=0.9)
    N = 20
    x0, U, kp, X = reference_rollouts()
    n = len(X)
    u_prev = U[:, :, 0] + np.stack([rng.uniform(-0.12, 0.12, n), rng.uniform(-0.09, 0.09, n)], axis=1)
    keep = rng.random(n) < 0.4                             # many exactly rate-feasible first steps
    u_prev[keep] = U[keep, :, 0]
    rolled = np.array([verdict_bits(mpc, X[i], U[i], u_prev[i], A, b, N) for i in range(n)], dtype=np.int32)

    def base():
        Xb = np.zeros((7, N + 1))
        Xb[5] = 2.0
        return Xb, np.zeros((2, N)), np.zeros(2)
    cases, names = [], []

    def add(name, Xe, Ue, up):
        names.append(name)
        cases.append((Xe, Ue, up))
    Xe, Ue, up = base(); add('all_inside', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[5, N] = 7.0; add('vN_above_5_accepted', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[5, N - 1] = 5.0; Ue[0, N - 1] = 0.0; add('v_exactly_5', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[5, 3] = np.nextafter(5.0, 6.0); add('v_one_ulp_above_5', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[5, 0] = 0.0; add('v_exactly_0', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[5, 7] = -1e-12; add('v_below_0', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[3, N] = 0.21; add('eyN_rejected', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[3, 0] = -0.2; Xe[3, N] = 0.2; add('ey_exactly_on_limits', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[3, 0] = np.nextafter(-0.2, -1.0); add('ey0_one_ulp_outside', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[0, :] = 3.0; up = np.array([3.0, 0.0]); Xe[5] = 0.5; add('a_exactly_max', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[0, :] = -4.0; up = np.array([-4.0, 0.0]); add('a_exactly_min_terminal_fails', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[0, 5:] = np.nextafter(3.0, 4.0); Ue[0, :5] = 3.0; up = np.array([3.0, 0.0]); add('a_one_ulp_above', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[1, :] = 1.0; up = np.array([0.0, 1.0]); add('df_exactly_max', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[1, :] = -1.0 - 1e-15; up = np.array([0.0, -1.0]); add('df_below_min', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[0] = 0.1 * 0.9 * np.arange(1, N + 1); Xe[5] = 0.2; add('rate_a_exactly_dt_jerk', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[1] = -0.1 * 0.7 * np.minimum(np.arange(1, N + 1), 10); add('rate_df_exactly', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[0, 0] = 0.0900001; add('rate_first_step_vs_u_prev', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[0, :] = 0.1 * 0.9; add('rate_first_step_exactly_dt_jerk_accepted', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[1, :] = -(0.1 * 0.7); add('rate_first_df_step_exactly_accepted', Xe, Ue, up)
    Xe, Ue, up = base(); Ue[1, 11:] = 0.0701; add('rate_df_mid_horizon', Xe, Ue, up)
    # terminal set: the pair is (v_{N-1}, a_{N-1}) -- a pair that is outside C_inf placed at N-1 fails, at N-2 or in x_N not
    ramp = np.maximum(0.0, 0.08 * (np.arange(N) - (N - 13)))          # 0 ... 0.96 at k = N-1, 0.08 per step (< dt*jerk)
    Xe, Ue, up = base(); Xe[5, N - 1] = 4.9; Ue[0] = ramp; add('terminal_outside_at_Nm1', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[5, N - 2] = 4.9; Ue[0, :N - 1] = ramp[1:]; Ue[0, N - 1] = 0.88; add('terminal_pair_at_Nm2_accepted', Xe, Ue, up)
    Xe, Ue, up = base(); Xe[5, N] = 4.99; Ue[0, N - 1] = 0.05; add('terminal_ignores_xN', Xe, Ue, up)
    for j in range(24):      # random trajectories around the limits
        Xe, Ue, up = base()
        Xe[5] = rng.uniform(-0.2, 5.3, N + 1)
        Xe[3] = rng.uniform(-0.22, 0.22, N + 1)
        Ue[0] = np.cumsum(rng.uniform(-0.095, 0.095, N))
        Ue[1] = np.cumsum(rng.uniform(-0.073, 0.073, N))
        add(f'random_{j}', Xe, Ue, up)
    eX = np.stack([c[0] for c in cases]); eU = np.stack([c[1] for c in cases]); eup = np.stack([c[2] for c in cases])
    edge = np.array([verdict_bits(mpc, eX[i], eU[i], eup[i], A, b, N) for i in range(len(cases))], dtype=np.int32)
    return dict(rolled_u_prev=u_prev, rolled_bits=rolled, edge_X=eX, edge_U=eU, edge_u_prev=eup, edge_bits=edge,
                edge_names=np.array(names), cinf_A=A, cinf_b=b)


ALL_ROUTES = ['12', '13', '14', '21', '23', '24', '31', '32', '34', '41', '42', '43']


def scenario_fixture(utils):
    """utils.py:141-169 on every ordered route pair (132): the 64 that belong to a scenario -> [e0, e1], the others
    raise 'Scenario not found'; utils.py:393-402 route_encoding."""
    enc = {}
    for r0 in ALL_ROUTES:
        for r1 in ALL_ROUTES:
            if r0 == r1:
                continue
            try:
                enc[f'{r0},{r1}'] = [int(v) for v in utils['scenario_encoding']([r0, r1])]
            except ValueError:
                enc[f'{r0},{r1}'] = None
    return dict(scenario_encoding=enc, route_encoding=dict(zip(ALL_ROUTES, utils['route_encoding'](ALL_ROUTES))))


def filter_fixture(utils, R, rng, n=500, N=20):
    """utils.py:365-388 on random two-vehicle scenes (+ scenes with the obstacle exactly abeam: dot == 0 is kept)."""
    ego = np.column_stack([rng.uniform(0, 50, n), rng.uniform(-20, 30, n), rng.uniform(-np.pi, np.pi, n)])
    obs = np.stack([rng.uniform(0, 50, (n, N + 1)), rng.uniform(-20, 30, (n, N + 1))], axis=1)
    for i in range(0, 40, 2):         # abeam / just behind / just ahead
        h = ego[i, 2] = [0.0, np.pi / 2, -np.pi / 2, np.pi][(i // 2) % 4]
        side = np.array([-np.sin(h), np.cos(h)]) * rng.uniform(1, 8)
        along = np.array([np.cos(h), np.sin(h)]) * [0.0, 1e-9, -1e-9][(i // 8) % 3]
        obs[i, :, 0] = ego[i, :2] + side + along
    out = np.zeros_like(obs)
    mk = lambda x, y, h: R['Ref']({'x': x, 'y': y, 'heading': h, 'v': 1.0, 's': 0.0, 'K': None, 'ey': 0, 'epsi': 0})
    for i in range(n):
        for ego_id in (0, 1):         # the function addresses the ego by index: exercise both
            preds = [None, None]
            preds[ego_id] = [mk(ego[i, 0], ego[i, 1], ego[i, 2]) for _ in range(N + 1)]
            preds[1 - ego_id] = [mk(obs[i, 0, k], obs[i, 1, k], 0.3) for k in range(N + 1)]
            f = utils['filter_preds'](preds, ego_id)
            got = np.array([[p.x for p in f[1 - ego_id]], [p.y for p in f[1 - ego_id]]])
            if ego_id == 0:
                out[i] = got
            else:
                assert np.array_equal(got, out[i])
            assert preds[1 - ego_id][0].x == obs[i, 0, 0]              # the input is not modified (deepcopy, :366)
    return dict(ego_xyh=ego, obs_xy=obs, filtered_xy=out)


def augment_fixture(utils, R, rng, n=200, N=20):
    """utils.py:354-363 with the reference's numpy RK4 model (evaluate.py:193, 443) on previous solutions that are
    golden rollouts, edited so that the last step ends above v = 5 with a > 0 (retry with a = 0), still above 5 after
    the retry (clip to 5) and below -1 (clip to -1)."""
    x0, U, kp, X = reference_rollouts()
    model = R['Frenet'](2.235, 2.235, 2.0, 0.1, discretization='rk4', mode='numpy', num_rk4_steps=4)
    idx = rng.choice(len(X), n, replace=False)
    Xp, Up, kps = X[idx].copy(), U[idx].copy(), kp[idx]
    for i in range(n):
        kind = i % 5
        if kind == 1:                 # retry: v_N just under 5, a_last > 0
            Xp[i, 5, -1] = 5.0 - rng.uniform(0, 0.05); Up[i, 0, -1] = rng.uniform(0.6, 2.0)
        elif kind == 2:               # retry is not enough: v_N itself above 5 -> clipped
            Xp[i, 5, -1] = 5.0 + rng.uniform(0.01, 0.4); Up[i, 0, -1] = rng.uniform(0.1, 1.0)
        elif kind == 3:               # below -1 -> clipped to -1
            Xp[i, 5, -1] = -1.0 - rng.uniform(0.0, 0.3); Up[i, 0, -1] = -rng.uniform(0.5, 3.0)
    Xa, Ua = np.zeros_like(Xp), np.zeros_like(Up)
    for i in range(n):
        Xa[i], Ua[i] = utils['augment_prev_sol']((Xp[i], Up[i]), model, make_K(*kps[i]))
    return dict(x_sol_prev=Xp, u_sol_prev=Up, kp=kps, x_aug=Xa, u_aug=Ua)


def marshalling_fixture(mpc, R, rng, n=48, N=20):
    """mpc.py:280-294 update_initial_condition and 241-278 update_predictions with the set_value sink: which number lands
    in which parameter slot (x0 with |heading| on routes '32' / '41', the obstacle block, raw_preds, raw_preds_np)."""
    Ref = R['Ref']
    pairs = [('13', '23'), ('32', '42'), ('21', '41'), ('12', '32'), ('41', '34'), ('24', '41')]
    recs = dict(routes=[], ind=[], state=[], u_prev=[], preds=[], x0_param=[], u_prev_param=[], preds_param=[],
                raw_param=[], raw_np=[])
    for c in range(n):
        routes = list(pairs[c % len(pairs)])
        ind = c % 2
        st = rng.normal(size=7) * np.array([20, 20, 20, 0.1, 0.1, 2, 1])
        st[6] = -abs(st[6]) - 0.1 if c % 3 else st[6]                 # negative headings matter on '32' / '41'
        up = rng.uniform(-1, 1, 2)
        P = rng.normal(size=(2, N + 1, 7)) * np.array([20, 20, 20, 0.1, 0.1, 2, 2])
        me = planner_self(N=N, routes=routes, ind=ind, M=2, x0=ParamKeys('x0'), u_prev=ParamKeys('u_prev'),
                          preds=ParamKeys('preds'), raw_preds=ParamKeys('raw_preds'))
        agent = {'type': 'CAV', 'state': Ref(dict(zip(('x', 'y', 's', 'ey', 'epsi', 'v', 'heading'), st), K=None))}
        mpc['update_initial_condition'](me, agent, R['Act']({'a': up[0], 'df': up[1]}))
        V = me.opti.values
        x0p = np.array([V[('x0', i)] for i in range(7)])
        upp = np.array(V[ParamKeys('u_prev')], dtype=np.float64)
        me.opti = Sink()
        preds = [[Ref(dict(zip(('x', 'y', 's', 'ey', 'epsi', 'v', 'heading'), P[m, k]), K=None)) for k in range(N + 1)]
                 for m in range(2)]
        mpc['update_predictions'](me, preds, raw_preds=preds)
        V = me.opti.values
        block = np.array([[V[('preds', (r, k))] for k in range(N + 1)] for r in range(7)])
        raw = np.array([V[('raw_preds', ('all', j))] for j in range(14)])
        assert me.pred_ind == [1 - ind] and me.NN_query_time == -1
        for k, v in (('routes', routes), ('ind', ind), ('state', st), ('u_prev', up), ('preds', P), ('x0_param', x0p),
                     ('u_prev_param', upp), ('preds_param', block), ('raw_param', raw), ('raw_np', me.raw_preds_np[0])):
            recs[k].append(v)
    return {k: np.array(v) for k, v in recs.items()}


def value_features_fixture(mpc, utils, rng, n=64, N=20):
    """mpc.py:326-354 get_xN(mode='numpy') with include_route False (what the shipped configs say, sc*_config.yaml:8)
    and True: x_N = [s_tv, v_tv, e_tv, s_N - s_tv, v_N - v_tv, e_ego - e_tv]."""
    pairs = [k.split(',') for k, v in scenario_fixture(utils)['scenario_encoding'].items() if v is not None]
    out = dict(routes=[], ind=[], sN_vN=[], raw_np=[], xN_scenario=[], xN_route=[])
    for c in range(n):
        routes = pairs[(c * 7) % len(pairs)]
        ind = c % 2
        X = rng.normal(size=(7, N + 1)) * 10
        raw = rng.normal(size=(1, 14)) * 10
        me = planner_self(N=N, routes=routes, ind=ind, pred_ind=[1 - ind], raw_preds_np=raw)
        out['routes'].append(routes); out['ind'].append(ind); out['sN_vN'].append([X[2, -1], X[5, -1]])
        out['raw_np'].append(raw[0])
        out['xN_scenario'].append(mpc['get_xN'](me, X, mode='numpy', include_route=False)[:, 0])
        out['xN_route'].append(mpc['get_xN'](me, X, mode='numpy', include_route=True)[:, 0])
    return {k: np.array(v) for k, v in out.items()}


def frenet2global_straight_fixture(utils, consts_ref):
    """utils.py:532-553: the straight routes never reach a casadi call.  `ref` carries what ReferenceGen produced."""
    s = np.linspace(0.0, 60.0, 121)
    out = {'s': s}
    for route in ('13', '24', '31', '42'):
        k = consts_ref[route]
        ref = {'x': np.array([k['x0'], k['xN']]), 'y': np.array([k['y0'], k['yN']]), 'K': np.zeros(2)}
        out[route] = np.array([utils['frenet2global'](si, ref, route, 50, 11.4, 2.8)[:, 0] for si in s])
    return out


def extracted_fixtures(R, consts_ref):
    rng = np.random.default_rng(2027)            # its own stream: the round-1 fixtures above stay bit-identical
    utils, mpc = reference_functions(R)
    np.savez_compressed(os.path.join(HERE, 'cost_golden.npz'), **cost_fixture(mpc, rng))
    np.savez_compressed(os.path.join(HERE, 'verdict_golden.npz'), **verdict_fixture(mpc, rng))
    with open(os.path.join(HERE, 'scenario_encoding.json'), 'w') as f:
        json.dump(scenario_fixture(utils), f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, 'filter_preds_golden.npz'), **filter_fixture(utils, R, rng))
    np.savez_compressed(os.path.join(HERE, 'augment_prev_sol_golden.npz'), **augment_fixture(utils, R, rng))
    np.savez_compressed(os.path.join(HERE, 'marshalling_golden.npz'), **marshalling_fixture(mpc, R, rng))
    np.savez_compressed(os.path.join(HERE, 'value_features_golden.npz'), **value_features_fixture(mpc, utils, rng))
    np.savez_compressed(os.path.join(HERE, 'frenet2global_straight_golden.npz'),
                        **frenet2global_straight_fixture(utils, consts_ref))


def main():
    R = _import_reference()
    rng = np.random.default_rng(2026)
    x0, U, kp = frenet_cases(rng, 1024)
    X = run_frenet(R, x0, U, kp)
    np.savez_compressed(os.path.join(HERE, 'frenet_rk4_golden.npz'), x0=x0, U=U, kp=kp, X=X,
                        n_rk4=4, dt=0.1)
    # a second, smaller set with other discretisation parameters (planner default n_rk4=7, mpc.py:32)
    x0b, Ub, kpb = frenet_cases(rng, 96)
    Xb = run_frenet(R, x0b, Ub, kpb, n_rk4=7)
    np.savez_compressed(os.path.join(HERE, 'frenet_rk4_golden_rk7.npz'), x0=x0b, U=Ub, kp=kpb, X=Xb,
                        n_rk4=7, dt=0.1)
    z, u, zn = run_cartesian(R, rng, 256)
    np.savez_compressed(os.path.join(HERE, 'cartesian_euler_golden.npz'), z=z, u=u, z_next=zn)
    consts = route_constants(R)
    with open(os.path.join(HERE, 'route_constants.json'), 'w') as f:
        json.dump(consts, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, 'value_net_golden.npz'), **value_nets(rng))
    extracted_fixtures(R, consts)
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
