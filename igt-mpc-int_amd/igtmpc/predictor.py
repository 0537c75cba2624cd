"""Opponent motion forecasts.

PredictorBase keeps the reference's constructor (common/PredictorBase.py:5-7); `predict` has the
arity every call site actually uses (constant_acceleration_model.py:18: agents, agent_cur_inputs,
routes, refs -- the abstract stub's two-argument form is never called).  The shipped
PredictorBase.py cannot even be imported (it names a package that does not exist), so this is a
restatement of the interface, not of that file.

ConstantAccelerationModel restates constant_acceleration_model.py:18-82 over the route tables of
igtmpc.routes: constant acceleration in s (69-71), heading from the piecewise-linear psi_ref
(46-66), x,y from frenet2global (75).  Host-side numpy (setup of the per-step inputs, SURVEY
section 8f item 1 moves it onto the GPU)."""
import numpy as np

from . import routes as R
from .vehicle import VehicleReference


class PredictorBase:
    def __init__(self, N, dt):
        self.N = N      # prediction horizon
        self.dt = dt    # time step

    def predict(self, agents, agent_cur_inputs, routes, refs=None):
        raise NotImplementedError


class ConstantAccelerationModel(PredictorBase):
    def __init__(self, N=10, dt=0.1, constant_speed=False, v_min=-2.0, v_max=20.0):
        super().__init__(N, dt)
        self.constant_speed = constant_speed
        self.v_min, self.v_max = v_min, v_max          # fourwayint.yaml:23-24 (constant_acceleration_model.py:43-44)

    def predict_arrays(self, s0, v0, a, route_ids):
        """Vectorised core: s0,v0,a,route_ids [n] -> dict of [n, N+1] arrays (k = 0 is the current
        value of s and v; x,y,heading at k = 0 are filled by the caller from the true state)."""
        s0, v0, a = (np.asarray(q, dtype=np.float64) for q in (s0, v0, a))
        rid = np.asarray(route_ids)
        n = len(s0)
        out = {k: np.empty((n, self.N + 1)) for k in ('s', 'v', 'x', 'y', 'heading')}
        s, v = s0.copy(), v0.copy()
        for k in range(self.N + 1):
            if k > 0:
                s = s + (v * self.dt + 0.5 * a * self.dt ** 2)           # :70
                v = np.clip(v + a * self.dt, self.v_min, self.v_max)     # :71
            out['s'][:, k], out['v'][:, k] = s, v
            xy = R.frenet2global(rid, s)                                 # :75
            out['x'][:, k], out['y'][:, k] = xy[:, 0], xy[:, 1]
            out['heading'][:, k] = R.psi_ref(rid, s)                     # :74
        return out

    def predict(self, agents, agent_cur_inputs, routes, refs=None):
        """-> list[M] of list[N+1] of VehicleReference (ey = epsi = 0, K = None; :34-35, :80)."""
        M = len(agents)
        a = [0.0 if self.constant_speed else agent_cur_inputs[i].a for i in range(M)]      # :26-29
        st = [ag['state'] for ag in agents]
        arr = self.predict_arrays([q.s for q in st], [q.v for q in st], a, [R.ROUTE_ID[r] for r in routes])
        preds = []
        for i in range(M):
            row = [VehicleReference({'x': st[i].x, 'y': st[i].y, 'heading': st[i].heading, 'v': st[i].v,
                                     's': st[i].s, 'K': None, 'ey': 0, 'epsi': 0})]         # :40
            for k in range(1, self.N + 1):
                row.append(VehicleReference({'x': arr['x'][i, k], 'y': arr['y'][i, k], 'heading': arr['heading'][i, k],
                                             'v': arr['v'][i, k], 's': arr['s'][i, k], 'K': None, 'ey': 0, 'epsi': 0}))
            preds.append(row)
        return preds
