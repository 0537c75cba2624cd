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
placeholder module object (no attributes, nothing emulated).  Anything that would
actually call casadi (mpc.py, utils.py, constant_acceleration_model.py) is NOT run
and stays "parity unpinned" (see DESIGN.md).
"""
import json
import os
import sys
import types

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
    with open(os.path.join(HERE, 'route_constants.json'), 'w') as f:
        json.dump(route_constants(R), f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, 'value_net_golden.npz'), **value_nets(rng))
    print('golden fixtures written to', HERE)


if __name__ == '__main__':
    main()
