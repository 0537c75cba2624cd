#!/usr/bin/env python3
"""Developer probe (GPU box): what a control step costs on a straight route, inside an arc, and across an arc's break-points
-- the f64 search over batches made of one kind of scenario each (picked out of the benchmark generator's, 16 384 of each),
lattice candidates, terminal set on.      python tools/arc_cost_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import numpy as np
import torch
from igtmpc import BatchSolver
from igtmpc.cinf import cinf_halfplanes
from igtmpc.scenarios import make_batch

big = make_batch(262144, dtype=np.float64)
s0, v0, kp = big['x0'][:, 2], big['x0'][:, 5], big['kparams']
turn = kp[:, 2] != 0
reach = v0 * 2.0 + 1.0                                   # where the fastest candidate can get to in the horizon, roughly
kinds = {
    'straight route': ~turn,
    'inside the arc all horizon': turn & (s0 > kp[:, 0] + 1.0) & (s0 + reach < kp[:, 1] - 1.0),
    'enters the arc': turn & (s0 < kp[:, 0] - 0.5) & (s0 + reach > kp[:, 0] + 1.0),
    'leaves the arc': turn & (s0 > kp[:, 0]) & (s0 < kp[:, 1] - 0.5) & (s0 + reach > kp[:, 1] + 1.0),
    'before the arc, never reaches it': turn & (s0 + reach < kp[:, 0] - 1.0),
}
n = 16384
with BatchSolver(dtype='f64') as s:
    s.set_cinf(*cinf_halfplanes())
    s.set_profiling(True)
    for name, m in kinds.items():
        idx = np.nonzero(m)[0]
        if len(idx) < n:
            print(f'{name}: only {len(idx)} scenarios'); continue
        idx = idx[:n]
        args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
                for a in (big['x0'][idx], big['u_prev'][idx], big['kparams'][idx], big['flags'][idx], big['obs_xy'][idx])]
        out = s.solve(*args)
        ts = []
        for _ in range(5):
            s.solve(*args, out=out)
            torch.cuda.synchronize()
            ts.append(s.kernel_ms()[0])
        ok = float((out['status'] == 0).float().mean().item())
        print(f'{name:34s}: search {np.median(ts):7.3f} ms for {n} scenarios = {np.median(ts) / n * 1e6:6.1f} ns per scenario   solved {ok:.2f}', flush=True)
