#!/usr/bin/env python3
"""Analysis (CPU, oracle side): what would a derivative-based polish of the shooting winner buy?
The tracking family's winner u* is improved by `iters` rounds of: finite-difference gradient of the cost over the 2 N inputs
(2 N roll-outs), then a line search over 64 step lengths along the negative gradient, every trial projected onto the input box
and the rate limits by a sequential clamp and judged by the verdicts -- two 64-candidate units per round on the device.  Gap to
the SLSQP optimum of the same NLP (oracle/nlp_quality.py) before and after.      python tools/polish_probe.py [n=48]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import multiprocessing as mp
import numpy as np
import np_oracle as O
import nlp_quality as Q
from igtmpc.scenarios import make_batch
from igtmpc.cinf import cinf_halfplanes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
P = O.Params(); cinf = cinf_halfplanes()
b = {k: (np.asarray(v, dtype=np.float64) if v.dtype.kind == 'f' else v) for k, v in make_batch(n, dtype=np.float64).items()}
N = P.N
ra, rd = P.dt * P.jerk, P.dt * P.steer_rate

def project(U, u_prev):
    """sequential clamp to the box and the rate limits (a feasible control sequence near U); U [..., 2, N]"""
    out = np.empty_like(U)
    pa = np.broadcast_to(u_prev[..., 0], U.shape[:-2]).copy(); pd = np.broadcast_to(u_prev[..., 1], U.shape[:-2]).copy()
    for k in range(N):
        a = np.clip(np.clip(U[..., 0, k], pa - ra, pa + ra), P.a_min, P.a_max)
        d = np.clip(np.clip(U[..., 1, k], pd - rd, pd + rd), -P.df_max, P.df_max)
        out[..., 0, k] = a; out[..., 1, k] = d; pa, pd = a, d
    return out

def evalJ(i, U):
    x0 = O.apply_flags(b['x0'][i][None], b['flags'][i:i+1])[0]
    X = O.rollout_frenet(x0[None, :], U, b['kparams'][i][None, :], P)
    J = O.stage_cost(X, U, P)
    g, mask = O.constraint_violation(X, U, b['u_prev'][i][None, :], b['obs_xy'][i][None], cinf[0], cinf[1], P, check_rate=True)
    return J, (mask == 0) & np.isfinite(J), g

def polish_fd(i, u, iters, eps=1e-4, M=64):
    J0, f0, _ = evalJ(i, u[None]); J0 = J0[0]
    hist = [J0]
    for it in range(iters):
        Up = np.repeat(u[None], 2 * N, axis=0)
        for c in range(2 * N):
            Up[c, c // N, c % N] += eps
        Jp, _, _ = evalJ(i, Up)
        g = ((Jp - J0) / eps).reshape(2, N)
        d = -g
        # scale: the largest step moves some component by 4 rate limits
        scale = np.maximum(np.abs(d[0]).max() / (4 * ra), np.abs(d[1]).max() / (4 * rd)) + 1e-30
        d = d / scale
        al = 2.0 ** (-np.arange(M) / 3.0)                       # 1 .. 2^-21
        Uc = project(u[None] + al[:, None, None] * d[None], b['u_prev'][i])
        Jc, fc, _ = evalJ(i, Uc)
        Jc = np.where(fc, Jc, np.inf)
        m = Jc.argmin()
        if Jc[m] < J0:
            u, J0 = Uc[m], Jc[m]
        hist.append(J0)
    return u, J0, hist

def _slsqp(job):
    i, u0 = job
    r = Q.polish(b['x0'][i], b['u_prev'][i], b['kparams'][i], b['flags'][i], b['obs_xy'][i], cinf[0], cinf[1], P, u0, maxiter=100)
    return i, r['cost'], r['max_violation']

if __name__ == '__main__':
    args = (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'], *cinf, P)
    tr = O.solve_batch_refined(*args, refine_iters=0, cand='track')[0]
    idx = [i for i in range(n) if tr['status'][i] == 0]
    with mp.Pool(7) as pool:
        res = pool.map(_slsqp, [(i, tr['u'][i]) for i in idx], chunksize=1)
    Jopt = {i: c for i, c, v in res if v < 1e-6}
    rows = []
    for i in idx:
        if i not in Jopt: continue
        u, J, hist = polish_fd(i, tr['u'][i], 4)
        jo = min(Jopt[i], min(hist))
        rows.append([h - jo for h in hist])
    rows = np.array(rows)
    print('scenarios', len(rows))
    for it in range(rows.shape[1]):
        print(f'after {it} polish iterations: gap mean {rows[:, it].mean():.4f} median {np.median(rows[:, it]):.4f} p90 {np.quantile(rows[:, it], .9):.4f} max {rows[:, it].max():.4f}')
