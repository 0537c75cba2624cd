#!/usr/bin/env python3
"""Developer probe (GPU box): how much of the f32 entry's disagreement with the float64 oracle is explained by
(a) constraint thresholds, (b) curvature break-points, (c) cost near-ties -- as a function of the set-aside width."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('igt-mpc-int_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import np_oracle as O
from helpers import ambiguous_mask, oracle_params, rel_err
from igtmpc import BatchSolver
from igtmpc.cinf import cinf_halfplanes
from igtmpc.scenarios import make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = make_batch(B, dtype=np.float32)
cinf = cinf_halfplanes()
with BatchSolver(dtype='f32') as s:
    P = oracle_params(s)
    s.set_cinf(*cinf)
    got = s.solve(b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])
    allc = s.rollout_all(b['x0'][:256], b['u_prev'][:256], b['kparams'][:256], b['flags'][:256], b['obs_xy'][:256], want_U=False)
f = lambda k: np.asarray(b[k], dtype=np.float64)
ref = O.solve_batch(f('x0'), f('u_prev'), f('kparams'), b['flags'], f('obs_xy'), *cinf, P, return_all=True)
kp = f('kparams')[:, None, :]
x0 = O.apply_flags(f('x0'), b['flags'])[:, None, :]
bp = O.breakpoint_distance(x0, ref['U'], kp, P)
wrong = (got['argmin'] != ref['argmin']) | (got['status'] != ref['status'])
print(f'B={B}: arg-min/status differs from the oracle on {wrong.sum()} scenarios ({wrong.mean():.4%})')
for e_thr, e_bp, e_tie in ((2e-5, 2e-5, 2e-5), (1e-6, 1e-6, 2e-5), (1e-7, 1e-7, 2e-5), (1e-7, 1e-7, 1e-6), (1e-8, 1e-8, 1e-7), (0, 0, 1e-7), (0, 0, 0)):
    amb = ambiguous_mask(ref, P, e_thr, e_tie, e_bp, bp)
    unexplained = wrong & ~amb
    print(f'  set aside thr<{e_thr:g} bp<{e_bp:g} tie<{e_tie:g}: {amb.sum():5d} scenarios ({amb.mean():.4%}); '
          f'differences outside it: {unexplained.sum()}')
    for i in np.nonzero(unexplained)[0][:3]:
        c, r = got['argmin'][i], ref['argmin'][i]
        print(f'      b={i}: device {c} oracle {r}; oracle J[dev]={ref["J"][i, c] if c >= 0 else None}, J[ref]={ref["J"][i, r] if r >= 0 else None}, '
              f'g[dev]={ref["g"][i, c] if c >= 0 else None} g[ref]={ref["g"][i, r] if r >= 0 else None} bp[dev]={bp[i, c] if c >= 0 else None} bp[ref]={bp[i, r] if r >= 0 else None}')
# trajectories: error vs break-point distance, every candidate of 256 scenarios
err = rel_err(allc['X'], ref['X'][:256]).max(axis=(-1, -2))
d = bp[:256]
for lo, hi in ((0, 1e-9), (1e-9, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 2e-5), (2e-5, np.inf)):
    m = (d >= lo) & (d < hi)
    if m.any():
        print(f'  rollouts with min |s_stage - b| in [{lo:g}, {hi:g}): {m.sum():6d}, max rel err {err[m].max():.2e}, share > 1e-5: {(err[m] > 1e-5).mean():.3f}')
print(f'  share of all {err.size} rollouts above 1e-5: {(err > 1e-5).mean():.2e}')
