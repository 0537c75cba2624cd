#!/usr/bin/env python3
"""How far are the shooting solver's answers from a local optimum of the reference's NLP?  CPU only.
shooting = the float64 oracle (same candidates as the device); optimum = scipy SLSQP on a single-shooting restatement of
mpc.py:147-160 (oracle/nlp_quality.py), started from the best answer any family found for the scenario, so every family
is measured against ONE optimum per scenario:  gap = J_family - J_opt  (cost of mpc.py:356-373, lower is better).
    python tools/nlp_gap.py [n_scenarios=256] [processes=7]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import multiprocessing as mp
import numpy as np
import np_oracle as O
import nlp_quality as Q
from igtmpc.scenarios import make_batch
from igtmpc.cinf import cinf_halfplanes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
procs = int(sys.argv[2]) if len(sys.argv) > 2 else 7
P = O.Params()
cinf = cinf_halfplanes()
b = {k: (np.asarray(v, dtype=np.float64) if v.dtype.kind == 'f' else v) for k, v in make_batch(n, dtype=np.float64).items()}


def _polish(job):
    i, u0 = job
    r = Q.polish(b['x0'][i], b['u_prev'][i], b['kparams'][i], b['flags'][i], b['obs_xy'][i], cinf[0], cinf[1], P, u0, maxiter=100)
    return i, r['cost'], r['max_violation']


def main():
    args = (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'], *cinf, P)
    fam = {'lattice (SURVEY 8d v0)': O.solve_batch(*args)}
    rh = O.solve_batch_refined(*args, refine_iters=2)
    fam['ramp-hold'], fam['ramp-hold + 2 refinements'] = rh[0], rh[-1]
    fam['tracking, no envelope'] = O.solve_batch_refined(*args, cand='track', track=dict(env=0.0))[0]
    fam['tracking, no speed cap (rounds 2-3)'] = O.solve_batch_refined(*args, cand='track', track=dict(vcap=0.0))[0]
    tr = O.solve_batch_refined(*args, refine_iters=2, cand='track')
    fam['tracking (default)'], fam['tracking + 2 refinements'] = tr[0], tr[-1]
    J = np.stack([np.where(f['status'] == 0, f['cost'], np.inf) for f in fam.values()])      # [families, n]
    best = J.argmin(axis=0)
    jobs = [(i, list(fam.values())[best[i]]['u'][i]) for i in range(n) if np.isfinite(J[:, i].min())]
    with mp.Pool(procs) as pool:
        res = pool.map(_polish, jobs, chunksize=1)
    J_opt = np.full(n, np.nan)
    for i, c, viol in res:
        if viol < 1e-6:
            J_opt[i] = min(c, J[:, i].min())
    ok = np.isfinite(J_opt)
    print(f'{n} scenarios of the benchmark generator; {len(jobs)} solved by at least one family; optimum (SLSQP, violation < 1e-6) '
          f'for {ok.sum()}, mean J_opt {J_opt[ok].mean():.4f}')
    for (name, f), Jf in zip(fam.items(), J):
        m = ok & np.isfinite(Jf)
        g = Jf[m] - J_opt[m]
        print(f'{name:36s}: solves {np.isfinite(Jf).mean() * 100:5.1f} %   gap mean {g.mean():.4f}  median {np.median(g):.4f}  '
              f'p90 {np.quantile(g, 0.9):.4f}  max {g.max():.4f}   ({m.sum()} scenarios)')


if __name__ == '__main__':
    main()
