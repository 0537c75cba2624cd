#!/usr/bin/env python3
"""How far are the shooting solver's answers from a local optimum of the reference's NLP?  CPU only:
shooting = the float64 oracle (same candidates as the device), polish = scipy SLSQP (oracle/nlp_quality.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import np_oracle as O
import nlp_quality as Q
from igtmpc.scenarios import make_batch
from igtmpc.cinf import cinf_halfplanes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
P = O.Params()
cinf = cinf_halfplanes()
b = {k: (np.asarray(v, dtype=np.float64) if v.dtype.kind == 'f' else v) for k, v in make_batch(64, dtype=np.float64).items()}
fam = {'lattice (SURVEY 8d v0)': O.solve_batch(b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'], *cinf, P)}
passes = O.solve_batch_refined(b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'], *cinf, P, refine_iters=2)
fam['ramp-hold'] = passes[0]
fam['ramp-hold + 2 refinements'] = passes[-1]
tr = O.solve_batch_refined(b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'], *cinf, P, refine_iters=2, cand='track')
fam['tracking'] = tr[0]
fam['tracking + 2 refinements'] = tr[-1]
idx = [i for i in range(64) if all(f['status'][i] == 0 for f in fam.values())][:n]
print(f'{len(idx)} scenarios solvable by every family; cost = mpc.py:356-373 (lower is better)')
base = None
for name, sol in fam.items():
    rows = Q.gap_report(b, sol, cinf, P, idx)
    if base is None:
        base = rows
    print(f'{name:28s}: shooting cost mean {rows[:, 1].mean():8.4f}   after SLSQP polish {rows[:, 2].mean():8.4f}   '
          f'gap mean {rows[:, 3].mean():.4f}  median {np.median(rows[:, 3]):.4f}  max {rows[:, 3].max():.4f}   ({len(rows)} polished)')
