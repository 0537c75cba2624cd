#!/usr/bin/env python3
"""Developer probe (GPU box): kernel times of one solve per candidate family (the benchmark's lattice beside the ramp-hold
and tracking families, first pass only and with one refinement pass), both entry points.
    python tools/family_probe.py [B ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import torch  # noqa: E402

from igtmpc import BatchSolver  # noqa: E402
from igtmpc.cinf import cinf_halfplanes  # noqa: E402
from igtmpc.scenarios import make_batch  # noqa: E402

for B in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
    for dtype in ('f64', 'f32'):
        npdt = np.float64 if dtype == 'f64' else np.float32
        b = make_batch(B, dtype=npdt)
        args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
                for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
        for cand, ri in (('lattice', 0), ('ramp_hold', 0), ('track', 0), ('track', 1)):
            with BatchSolver(dtype=dtype, cand_mode=cand, refine_iters=ri) as s:
                s.set_cinf(*cinf_halfplanes())
                s.set_profiling(True)
                out = s.solve(*args)
                torch.cuda.synchronize()
                ts, te = [], []
                for _ in range(7):
                    s.solve(*args, out=out)
                    torch.cuda.synchronize()
                    a, e = s.kernel_ms()
                    ts.append(a); te.append(e)
                ok = float((out['status'] == 0).float().mean())
            ms, me = np.median(ts), np.median(te)
            print(f'{dtype} B={B:6d} {cand:9s} refine={ri}: search {ms:8.3f} ms  emit {me:7.3f} ms -> '
                  f'{B / (ms + me) * 1e3 / 1e6:7.3f} M solves/s (kernels)  solved {ok:.3f}', flush=True)
