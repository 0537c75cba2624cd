"""TEST INFRASTRUCTURE ONLY -- ctypes loader of oracle/liboracle.so (igt_oracle.c).
Used by tests (second opinion next to np_oracle) and by bench.py's cpu_baseline leg."""
import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle.so')


class orc_params(ct.Structure):
    _fields_ = [('N', ct.c_int32), ('n_rk4', ct.c_int32), ('C', ct.c_int32), ('n_obs', ct.c_int32),
                ('G', ct.c_int32), ('F', ct.c_int32),
                ('dt', ct.c_double), ('l_r', ct.c_double), ('l_f', ct.c_double),
                ('v_min', ct.c_double), ('v_max', ct.c_double), ('a_min', ct.c_double), ('a_max', ct.c_double),
                ('df_max', ct.c_double), ('jerk', ct.c_double), ('steer_rate', ct.c_double),
                ('ey_lim', ct.c_double), ('d_min', ct.c_double), ('w_u', ct.c_double), ('feas_tol', ct.c_double)]


_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) and build:
            subprocess.run(['make', '-C', _HERE], check=True)
        _lib = ct.CDLL(_SO)
        _lib.orc_max_threads.restype = ct.c_int
    return _lib


def _params(P, C, n_obs, F, table):
    G = 1 if table is not None else int(round(np.sqrt(C)))
    return orc_params(P.N, P.n_rk4, C, n_obs, G, F, P.dt, P.l_r, P.l_f, P.v_min, P.v_max, P.a_min, P.a_max,
                      P.df_max, P.jerk, P.steer_rate, P.ey_lim, P.d_min, P.w_u, P.feas_tol)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(ct.c_void_p)


def solve_batch(x0, u_prev, kp, flags, obs_xy, cinf_A, cinf_b, P, C=256, U=None, nthreads=1):
    lib = load()
    x0, u_prev, kp = _f(x0), _f(u_prev), _f(kp)
    B = x0.shape[0]
    obs = _f(obs_xy) if obs_xy is not None else np.zeros((B, 0, 2, P.N + 1))
    n_obs = obs.shape[1]
    A = _f(cinf_A) if cinf_A is not None else None
    b = _f(cinf_b) if cinf_b is not None else None
    F = 0 if b is None else len(b)
    tab = _f(U) if U is not None else None
    pr = _params(P, C, n_obs, F, tab)
    fl = np.ascontiguousarray(flags, dtype=np.uint32)
    out = dict(x=np.empty((B, 7, P.N + 1)), u=np.empty((B, 2, P.N)), cost=np.empty(B),
               argmin=np.empty(B, np.int32), status=np.empty(B, np.int32))
    lib.orc_solve_batch(ct.byref(pr), B, _p(x0), _p(u_prev), _p(kp), _p(fl), _p(obs), _p(A), _p(b), _p(tab),
                        _p(out['x']), _p(out['u']), _p(out['cost']), _p(out['argmin']), _p(out['status']),
                        int(nthreads))
    return out


def rollout_all(x0, u_prev, kp, flags, obs_xy, cinf_A, cinf_b, P, C=256, U=None, nthreads=1):
    lib = load()
    x0, u_prev, kp = _f(x0), _f(u_prev), _f(kp)
    B = x0.shape[0]
    obs = _f(obs_xy) if obs_xy is not None else np.zeros((B, 0, 2, P.N + 1))
    A = _f(cinf_A) if cinf_A is not None else None
    b = _f(cinf_b) if cinf_b is not None else None
    tab = _f(U) if U is not None else None
    pr = _params(P, C, obs.shape[1], 0 if b is None else len(b), tab)
    fl = np.ascontiguousarray(flags, dtype=np.uint32)
    out = dict(X=np.empty((B, C, 7, P.N + 1)), U=np.empty((B, C, 2, P.N)), cost=np.empty((B, C)),
               viol=np.empty((B, C), np.uint32))
    lib.orc_rollout_all(ct.byref(pr), B, _p(x0), _p(u_prev), _p(kp), _p(fl), _p(obs), _p(A), _p(b), _p(tab),
                        _p(out['X']), _p(out['U']), _p(out['cost']), _p(out['viol']), int(nthreads))
    return out


def max_threads():
    return load().orc_max_threads()
