"""TEST INFRASTRUCTURE ONLY -- quality yardstick, not a parity oracle.

Restates the reference's NLP (mpc.py:147-160: cost 356-373, constraints 177-180, 223-226, 296-321) in
single-shooting form over the 2N inputs and hands it to scipy.optimize (SLSQP) -- the closest stand-in for
CasADi/IPOPT that exists in this image -- to measure how far the shooting solver's best candidate is from a
local optimum of the same problem.  Started from the shooting solution, so "gap" = what a local NLP polish still
gains.  The states are eliminated by rolling the oracle's model (np_oracle.rollout_frenet), which is the
reference's own arithmetic (bit-exact against the golden vectors)."""
import numpy as np
from scipy.optimize import minimize

import np_oracle as O


def polish(x0, u_prev, kp, flag, obs_xy, cinf_A, cinf_b, P, u_init, maxiter=60):
    """One scenario.  u_init[2,N] (e.g. the shooting winner).  -> dict(u, cost, cost_init, max_violation, ok)."""
    N = P.N
    x0 = O.apply_flags(np.asarray(x0, dtype=np.float64)[None], np.asarray([flag]))[0]
    kp = np.asarray(kp, dtype=np.float64)
    ra, rd = P.dt * P.jerk, P.dt * P.steer_rate

    def unpack(z):
        return z.reshape(2, N)

    def roll(z):
        U = unpack(z)
        return O.rollout_frenet(x0, U, kp, P), U

    def cost(z):
        X, U = roll(z)
        return float(O.stage_cost(X, U, P))

    def ineq(z):          # scipy convention: every entry >= 0
        X, U = roll(z)
        v = X[O.IV, :N]
        a, d = U[0], U[1]
        a_prev = np.concatenate([[u_prev[0]], a[:-1]])
        d_prev = np.concatenate([[u_prev[1]], d[:-1]])
        g = [v - P.v_min, P.v_max - v,                                    # mpc.py:316-317
             ra - (a - a_prev), ra + (a - a_prev), rd - (d - d_prev), rd + (d - d_prev),   # mpc.py:301-312
             P.ey_lim - X[O.IEY], P.ey_lim + X[O.IEY]]                    # mpc.py:296-299
        if cinf_A is not None:
            g.append(cinf_b - (cinf_A[:, 0] * X[O.IV, N - 1] + cinf_A[:, 1] * a[N - 1]))   # mpc.py:177-180
        if obs_xy is not None and len(obs_xy):
            for o in obs_xy:
                g.append((X[O.IX, 1:] - o[0, 1:]) ** 2 + (X[O.IY, 1:] - o[1, 1:]) ** 2 - P.d_min ** 2)  # mpc.py:223-226
        return np.concatenate([np.atleast_1d(q).ravel() for q in g])

    z0 = np.asarray(u_init, dtype=np.float64).ravel()
    bounds = [(P.a_min, P.a_max)] * N + [(-P.df_max, P.df_max)] * N      # mpc.py:318-321
    res = minimize(cost, z0, method='SLSQP', bounds=bounds, constraints=[{'type': 'ineq', 'fun': ineq}],
                   options={'maxiter': maxiter, 'ftol': 1e-7})
    viol = float(np.maximum(-ineq(res.x), 0).max())
    return dict(u=unpack(res.x), cost=float(res.fun), cost_init=cost(z0), max_violation=viol,
                ok=bool(res.success or viol < 1e-5), nit=int(res.nit))


def gap_report(batch, sols, cinf, P, idx):
    """For scenarios `idx` with a shooting solution sols['u'][i]: polish and report the cost gap."""
    rows = []
    for i in idx:
        if sols['status'][i] != 0:
            continue
        r = polish(batch['x0'][i], batch['u_prev'][i], batch['kparams'][i], batch['flags'][i], batch['obs_xy'][i],
                   cinf[0], cinf[1], P, sols['u'][i])
        if r['max_violation'] < 1e-4:
            rows.append((i, r['cost_init'], r['cost'], r['cost_init'] - r['cost'], r['nit']))
    return np.array(rows)
