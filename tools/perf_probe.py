#!/usr/bin/env python3
"""Developer probe (GPU box): kernel times of the search/emit pair for a few configurations."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import torch  # noqa: E402

from igtmpc import BatchSolver  # noqa: E402
from igtmpc.scenarios import make_batch  # noqa: E402


def run(dtype, B, nc, iters=5, straight=None, net=None):
    os.environ['IGT_NC'] = str(nc)
    npdt = np.float32 if dtype == 'f32' else np.float64
    b = make_batch(B, dtype=npdt)
    if straight is True:
        b['kparams'][:] = (np.inf, np.inf, 0.0)
    elif straight is False:
        b['kparams'][:] = (19.3, 32.8088, 0.11627907)
    args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda()
            for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
    if net is not None:
        args += [torch.from_numpy(b['tv_sv']).cuda(), torch.from_numpy(b['enc']).cuda()]
    with BatchSolver(dtype=dtype, cost_mode='value_net' if net is not None else 'progress') as s:
        if net is not None:
            s.set_value_net(net)
        s.set_profiling(True)
        out = s.solve(*args)
        torch.cuda.synchronize()
        ts, te, wall = [], [], []
        for _ in range(iters):
            t0 = time.perf_counter()
            s.solve(*args, out=out)
            torch.cuda.synchronize()
            wall.append(time.perf_counter() - t0)
            a, e = s.kernel_ms()
            ts.append(a)
            te.append(e)
        st = out['status'].cpu().numpy()
    ms, me, mw = np.median(ts), np.median(te), np.median(wall) * 1e3
    print(f'{dtype} B={B:6d} straight={straight} net={None if net is None else len(net)}: search {ms:8.3f} ms  emit {me:7.3f} ms  wall {mw:8.3f} ms  '
          f'-> {B / (ms + me) * 1e3 / 1e6:7.3f} M solves/s (kernels)  feasible {np.mean(st == 0):.2f}', flush=True)


def host_mode(B=4096):
    b = make_batch(B, dtype=np.float32)
    a = (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])
    with BatchSolver(dtype='f32') as s:
        s.solve(*a)
        t0 = time.perf_counter()
        for _ in range(20):
            s.solve(*a)
        dt = (time.perf_counter() - t0) / 20
    print(f'host buffers (PCIe-inclusive, staged) B={B}: {dt * 1e3:.3f} ms/solve-call -> {B / dt / 1e6:.3f} M solves/s', flush=True)


if __name__ == '__main__':
    host_mode()
    run('f32', 4096, 2)
    run('f32', 65536, 2)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'value_net_golden.npz'))
    for sc in (1, 3):
        layers, i = [], 0
        while f'sc{sc}_W{i}' in g:
            layers.append((g[f'sc{sc}_W{i}'], g[f'sc{sc}_b{i}'])); i += 1
        run('f32', 65536, 2, net=layers)
        run('f32', 4096, 2, net=layers)
    if len(sys.argv) > 1:
        run('f64', 4096, 1, iters=2)
