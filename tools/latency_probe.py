#!/usr/bin/env python3
"""Developer probe (GPU box): latency of ONE solve at the batch sizes a per-agent caller has (B = 1, 2, 32), host buffers
(numpy in / numpy out, what MPC_Planner.solve does) and device buffers, tracking candidates, f64.
    python tools/latency_probe.py [B ...]        IGT_DEV_FLAGS=524288: without the trajectories kept by the search pass"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import torch  # noqa: E402

from igtmpc import BatchSolver  # noqa: E402
from igtmpc.cinf import cinf_halfplanes  # noqa: E402
from igtmpc.scenarios import make_batch  # noqa: E402

BATCHES = [int(a) for a in sys.argv[1:]] or [1, 2, 32]
for N in (20, 40):
    for B in BATCHES:
        b = make_batch(max(B, 8), N=N, dtype=np.float64)
        host = [np.ascontiguousarray(b[k][:B]) for k in ('x0', 'u_prev', 'kparams', 'flags', 'obs_xy')]
        dev = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in host]
        with BatchSolver(N=N, dtype='f64', cand_mode='track') as s:
            s.set_cinf(*cinf_halfplanes())
            row = {}
            for name, args in (('host', host), ('device', dev)):
                out = s.solve(*args)
                torch.cuda.synchronize()
                ts = []
                for _ in range(300):
                    t0 = time.perf_counter()
                    s.solve(*args, out=out if name == 'device' else None)
                    if name == 'device':
                        torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                row[name] = (np.median(ts) * 1e6, np.percentile(ts, 95) * 1e6)
            s.set_profiling(True)
            s.solve(*dev)
            torch.cuda.synchronize()
            k = s.kernel_ms()
        print(f'N={N} B={B:3d}: host buffers {row["host"][0]:7.1f} us (p95 {row["host"][1]:7.1f})   device buffers '
              f'{row["device"][0]:7.1f} us (p95 {row["device"][1]:7.1f})   kernels: search {k[0] * 1e3:6.1f} us, emit {k[1] * 1e3:5.1f} us', flush=True)
