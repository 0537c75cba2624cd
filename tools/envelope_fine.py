import os, sys
sys.path.insert(0, 'igt-mpc-int_amd')
import numpy as np
from igtmpc.evaluate import run_closed_loop
for N, envs in ((20, (0.35, 0.4, 0.45, 0.55, 0.6)), (40, (0.4, 0.6, 0.8))):
    for env in envs:
        inf, dl, s = [], [], []
        for sc in range(1, 9):
            r = run_closed_loop(sc=sc, num_samples=64, N=N, cand_mode='track', warm_start=False, dtype='f64', limits={'track_env': env})
            inf.append(r['infeasible_ratio'].mean()); dl.append(r['deadlock'].mean()); s.append(r['x_data'][:, 2::7, -1].mean())
        print(f'N={N} warm=0 track_env={env:.2f}: infeasible steps {np.mean(inf):.3f}  deadlock flag {np.mean(dl):.3f}  mean final s {np.mean(s):.1f} m', flush=True)
