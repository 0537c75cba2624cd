#!/usr/bin/env python3
"""Developer probe (GPU box): the float64 entry point -- kernel times at a few batch sizes, the oracle-order kernels
(IGT_DEV_FLAGS=1024) beside the production ones, and how far the two are apart.
    python tools/f64_probe.py [B ...]"""
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


def run(B, dev_flags, iters=5, dtype='f64'):
    if dev_flags:
        os.environ['IGT_DEV_FLAGS'] = str(dev_flags)
    else:
        os.environ.pop('IGT_DEV_FLAGS', None)
    npdt = np.float64 if dtype == 'f64' else np.float32
    b = make_batch(B, dtype=npdt)
    args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
            for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
    with BatchSolver(dtype=dtype) as s:
        s.set_cinf(*cinf_halfplanes())
        s.set_profiling(True)
        out = s.solve(*args)
        torch.cuda.synchronize()
        ts, te = [], []
        for _ in range(iters):
            s.solve(*args, out=out)
            torch.cuda.synchronize()
            a, e = s.kernel_ms()
            ts.append(a)
            te.append(e)
        res = {k: v.cpu().numpy() for k, v in out.items()}
    ms, me = np.median(ts), np.median(te)
    print(f'{dtype} B={B:6d} dev={dev_flags:5d}: search {ms:8.3f} ms  emit {me:7.3f} ms -> {B / (ms + me) * 1e3 / 1e6:7.3f} M solves/s '
          f'(kernels)  feasible {np.mean(res["status"] == 0):.3f}', flush=True)
    return res


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1].startswith('--flags='):        # A/B of developer switches: --flags=0,32,16 B...
        flags = [int(f) for f in sys.argv[1][8:].split(',')]
        for B in [int(a) for a in sys.argv[2:]] or [4096]:
            for f in flags:
                run(B, f, iters=9, dtype=os.environ.get('IGT_PROBE_DTYPE', 'f64'))
        sys.exit(0)
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 32768]
    for B in sizes:
        fast = run(B, 0)
        if B <= 8192:
            ref = run(B, 1024, iters=2)
            same = fast['argmin'] == ref['argmin']
            ok = same & (ref['status'] == 0)
            ex = np.abs(fast['x'][ok] - ref['x'][ok]) / np.maximum(1, np.abs(ref['x'][ok]))
            ec = np.abs(fast['cost'][ok] - ref['cost'][ok]) / np.maximum(1, np.abs(ref['cost'][ok]))
            print(f'   production vs oracle-order kernels: arg-min equal {same.mean():.5f}, max rel dx {ex.max():.2e}, '
                  f'max rel dcost {ec.max():.2e}', flush=True)
        run(B, 0, dtype='f32')
        if B <= 4096:     # the literal north_star mapping (one wave per trajectory, stages in LDS), measurement variant
            lit = run(B, 2048, iters=2)
            same = fast['argmin'] == lit['argmin']
            print(f'   literal wave-per-trajectory variant: arg-min equal to production on {same.mean():.5f} of the scenarios', flush=True)
