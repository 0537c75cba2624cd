#!/usr/bin/env python3
"""Developer probe (GPU box): closed-loop statistics of the tracking family against the scale of its acceleration
envelope (igt_params.track_env; 0 = no envelope), at the benchmark's horizon and at the one the reference ships.
64 episodes x 150 steps (IGT_CL_EPISODES) per scenario, warm start on (IGT_CL_WARM=0: off), f64.      python tools/envelope_sweep.py [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import numpy as np
from igtmpc.evaluate import run_closed_loop
EPISODES = int(os.environ.get('IGT_CL_EPISODES', '64'))        # episodes per scenario
WARM = os.environ.get('IGT_CL_WARM', '1') != '0'
horizons = [int(a) for a in sys.argv[1:]] or [20, 40]
for N in horizons:
    for env in (0.0, 0.3, 0.5, 0.7, 1.0, 1.3):
        inf, dl, s, first = [], [], [], []
        for sc in range(1, 9):
            r = run_closed_loop(sc=sc, num_samples=EPISODES, N=N, cand_mode='track', warm_start=WARM, dtype='f64',
                                limits={'track_env': env})
            inf.append(r['infeasible_ratio'].mean()); dl.append(r['deadlock'].mean()); s.append(r['x_data'][:, 2::7, -1].mean())
        print(f'N={N} warm={int(WARM)} track_env={env:.1f}: infeasible steps {np.mean(inf):.3f}  deadlock flag {np.mean(dl):.3f}  '
              f'mean final s {np.mean(s):.1f} m', flush=True)
