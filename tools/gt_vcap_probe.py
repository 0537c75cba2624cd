#!/usr/bin/env python3
"""Developer probe (GPU box): closed loop with the gt_mpc cost (shipped value networks, identity statistics), tracking family,
with and without the speed cap of the acceleration targets (igt_params.track_vcap).  64 episodes x 150 steps per scenario."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import numpy as np
from igtmpc.evaluate import run_closed_loop
for N in (20, 40):
    for vcap in (1.0, 0.0):
        inf, dl, s = [], [], []
        for sc in range(1, 9):
            r = run_closed_loop(sc=sc, num_samples=64, N=N, cand_mode='track', eval_mode='gt_mpc', dtype='f64', limits={'track_vcap': vcap})
            inf.append(r['infeasible_ratio'].mean()); dl.append(r['deadlock'].mean()); s.append(r['x_data'][:, 2::7, -1].mean())
        print(f'gt_mpc N={N} track_vcap={vcap:.0f}: infeasible steps {np.mean(inf):.3f}  deadlock flag {np.mean(dl):.3f}  mean final s {np.mean(s):.1f} m', flush=True)
