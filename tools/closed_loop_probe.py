#!/usr/bin/env python3
"""Developer probe (GPU box): closed-loop statistics per scenario and candidate family (64 episodes x 150 steps (IGT_CL_EPISODES) each),
with and without the warm start (previous solution shifted by one step as the centre of the ramp-hold candidates)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import numpy as np
from igtmpc.evaluate import run_closed_loop
EPISODES = int(os.environ.get('IGT_CL_EPISODES', '64'))        # episodes per scenario
import itertools
dtype = sys.argv[1] if len(sys.argv) > 1 else 'f64'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20          # horizon (mpc.yaml:6 ships 40; BASELINE fixes 20)
cfgs = [('lattice', 0, False, None), ('ramp_hold', 0, False, None), ('ramp_hold', 0, True, None), ('ramp_hold', 1, True, None),
        ('ramp_hold', 0, True, 1e-3), ('track', 0, False, None), ('track', 0, True, None), ('track', 1, True, None),
        ('track', 0, True, 1e-3), ('track', 0, True, None, {'track_env': 0.0}), ('track', 0, True, None, {'track_env': 1.0})]
# ('track', ..., None) runs with the driver's horizon-aware envelope scale (igtmpc.evaluate.auto_track_env: 0.5 at N = 20, 1 at N = 40)
if len(sys.argv) > 3:                                       # only the configurations of one family
    cfgs = [c for c in cfgs if c[0] == sys.argv[3]]
tot = {}
for cfg, sc in itertools.product(cfgs, range(1, 9)):
    (cm, ri, ws, ft), lim = cfg[:4], (cfg[4] if len(cfg) > 4 else None)
    r = run_closed_loop(sc=sc, num_samples=EPISODES, N=N, cand_mode=cm, refine_iters=ri, warm_start=ws, dtype=dtype, feas_tol=ft, limits=lim)
    row = {'cand': cm, 'limits': lim, 'refine': ri, 'warm': ws, 'feas_tol': ft, 'sc': sc, 'infeasible': r['infeasible_ratio'].mean(axis=0).round(3).tolist(),
           'deadlock': float(r['deadlock'].mean()), 'final_s': r['x_data'][:, 2::7, -1].mean(axis=0).round(1).tolist(),
           'max|ey|': float(np.abs(r['x_data'][:, 3::7, :]).max().round(3)),
           'min_dist': float(np.hypot(r['x_data'][:, 0, :] - r['x_data'][:, 7, :], r['x_data'][:, 1, :] - r['x_data'][:, 8, :]).min().round(2)),
           'ms/step': float(r['solve_ms'][5:].mean().round(3))}
    print(json.dumps(row), flush=True)
    t = tot.setdefault((cm, ri, ws, ft, json.dumps(lim)), dict(inf=[], dl=[], s=[]))
    t['inf'].append(r['infeasible_ratio'].mean()); t['dl'].append(r['deadlock'].mean()); t['s'].append(r['x_data'][:, 2::7, -1].mean())
for k, t in tot.items():
    print(f'SUMMARY N={N} cand={k[0]} refine={k[1]} warm={k[2]} feas_tol={k[3]} limits={k[4]}: infeasible steps {np.mean(t["inf"]):.3f}  deadlock flag {np.mean(t["dl"]):.3f}  '
          f'mean final s {np.mean(t["s"]):.1f}', flush=True)
