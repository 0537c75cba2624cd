#!/usr/bin/env python3
"""Developer probe for rocprofv3 --pmc runs: search kernel on straight / curved batches at n_rk4 = 4 and 8."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import torch
from igtmpc import BatchSolver
from igtmpc.scenarios import make_batch
B = 16384
for n_rk4 in (4, 8):
    for mode in ('straight', 'curved', 'mix'):
        b = make_batch(B, dtype=np.float32)
        if mode == 'straight': b['kparams'][:] = (np.inf, np.inf, 0.0)
        if mode == 'curved': b['kparams'][:] = (-100.0, 1000.0, 0.11627907)
        args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
        with BatchSolver(dtype='f32', n_rk4=n_rk4) as s:
            out = s.solve(*args)
            torch.cuda.synchronize()
            print(n_rk4, mode, 'mean |ey_N| of winners', float(out['x'][:, 3, -1].nan_to_num().abs().mean()), 'feasible', float((out['status'] == 0).float().mean()))
