#!/usr/bin/env python3
"""Developer probe (GPU box): closed-loop statistics per scenario."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import numpy as np
from igtmpc.evaluate import run_closed_loop
import itertools
cfgs = [('lattice', 0), ('ramp_hold', 0), ('ramp_hold', 1), ('ramp_hold', 2)]
for (cm, ri), sc in itertools.product(cfgs, range(1, 9)):
    r = run_closed_loop(sc=sc, num_samples=16, N=20, cand_mode=cm, refine_iters=ri)
    print(json.dumps({'cand': cm, 'refine': ri, 'sc': sc, 'infeasible': r['infeasible_ratio'].mean(axis=0).round(3).tolist(),
                      'deadlock': float(r['deadlock'].mean()), 'final_s': r['x_data'][:, 2::7, -1].mean(axis=0).round(1).tolist(),
                      'max|ey|': float(np.abs(r['x_data'][:, 3::7, :]).max().round(3)),
                      'min_dist': float(np.hypot(r['x_data'][:, 0, :] - r['x_data'][:, 7, :], r['x_data'][:, 1, :] - r['x_data'][:, 8, :]).min().round(2)),
                      'ms/step': float(r['solve_ms'][5:].mean().round(3))}), flush=True)
