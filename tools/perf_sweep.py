#!/usr/bin/env python3
"""Developer probe: search-kernel time vs n_rk4 / N (separates per-step overhead from sub-step cost)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import torch
from igtmpc import BatchSolver
from igtmpc.scenarios import make_batch

def run(B, n_rk4, N, straight):
    b = make_batch(B, N=N, dtype=np.float32)
    if straight is True: b['kparams'][:] = (np.inf, np.inf, 0.0)
    elif straight is False: b['kparams'][:] = (-100.0, 1000.0, 0.11627907)
    args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
    with BatchSolver(dtype='f32', n_rk4=n_rk4, N=N) as s:
        s.set_profiling(True)
        out = s.solve(*args); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            s.solve(*args, out=out); torch.cuda.synchronize(); ts.append(s.kernel_ms()[0])
    print(f'B={B} N={N} n_rk4={n_rk4} straight={straight}: search {np.median(ts):.3f} ms', flush=True)

for st in (True, False):
    for n in (4, 8):
        run(65536, n, 20, st)
