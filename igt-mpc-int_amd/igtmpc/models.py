"""The reference's callable vehicle models, kept as objects: model(state, action) -> VehicleReference
(call sites utils.py:358 augment_prev_sol, evaluate.py:520 brake fallback, ReferenceGen.py:131).
The arithmetic runs in the HIP library (igt_frenet_step_f64 / igt_cartesian_euler_f64): there is no
CPU implementation in the product."""
import numpy as np

from .solver import BatchSolver
from .vehicle import Curvature, VehicleReference

_SOLVERS = {}


def _solver(dt, n_rk4, l_r, l_f, device):
    key = (dt, n_rk4, l_r, l_f, device)
    if key not in _SOLVERS:
        _SOLVERS[key] = BatchSolver(N=1, dt=dt, n_rk4=n_rk4, C=64, n_obs=0, device=device, dtype='f64',
                                    l_r=l_r, l_f=l_f)
    return _SOLVERS[key]


def _kparams(K):
    if K is None or K == 0:
        return Curvature().kparams
    if hasattr(K, 'kparams'):
        return K.kparams
    if callable(K):     # any K(s) of the reference's shape (evaluate.py:384-402): probed once for (b0, b1, Kv)
        return Curvature.from_callable(K).kparams
    raise TypeError('state.K must be None, 0, an igtmpc.Curvature or a callable piecewise-constant K(s)')


class KinematicBicycleModelFrenet:
    """RK4 Frenet bicycle, one control step (kinematic_bicycle_model_frenet.py:7-14 ctor, 70-127 step)."""

    def __init__(self, l_r, l_f, width, dt, discretization='rk4', mode='numpy', num_rk4_steps=10, device=0):
        if discretization != 'rk4':
            # the reference's numpy Euler branch is broken (UnboundLocalError, lines 38-40) and its
            # casadi Euler branch is never used by the planner
            raise ValueError('only discretization="rk4" is part of the path')
        self.l_r, self.l_f, self.width, self.delta_t = l_r, l_f, width, dt
        self.discretization, self.mode, self.num_rk4_steps = discretization, mode, num_rk4_steps
        self._s = _solver(dt, num_rk4_steps, l_r, l_f, device)

    def __call__(self, state, action):
        x = np.array([[state.x, state.y, state.s, state.ey, state.epsi, state.v, state.heading]], dtype=np.float64)
        u = np.array([[action.a, action.df]], dtype=np.float64)
        kp = np.array([_kparams(state.K)], dtype=np.float64)
        o = self._s.frenet_step(x, u, kp)[0]
        return VehicleReference({'x': o[0], 'y': o[1], 's': o[2], 'ey': o[3], 'epsi': o[4], 'v': o[5],
                                 'heading': o[6], 'K': state.K})

    def step_batch(self, x, u, kparams):
        """x[n,7], u[n,2], kparams[n,3] -> x_next[n,7] (numpy or CUDA tensors)."""
        return self._s.frenet_step(x, u, kparams)


class KinematicBicycleModel:
    """4-state Cartesian forward Euler (kinematic_bicycle_model.py:7-13 ctor, 15-50 step)."""

    def __init__(self, l_r, l_f, width, dt, discretization='euler', mode='numpy', device=0):
        if discretization != 'euler':
            raise NotImplementedError           # as the reference (kinematic_bicycle_model.py:46-47)
        self.l_r, self.l_f, self.width, self.delta_t = l_r, l_f, width, dt
        self.discretization, self.mode = discretization, mode
        self._s = _solver(dt, 1, l_r, l_f, device)

    def __call__(self, state, action):
        z = np.array([[state.x, state.y, state.heading, state.v]], dtype=np.float64)
        u = np.array([[[action.a], [action.df]]], dtype=np.float64)
        o = self._s.cartesian_euler(z, u)[0, :, 1]
        return VehicleReference({'x': o[0], 'y': o[1], 'heading': o[2], 'v': o[3], 's': 0, 'ey': 0, 'epsi': 0, 'K': 0})
