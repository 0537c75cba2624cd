#!/usr/bin/env python3
"""Developer probe (GPU box): timeline of the search kernel's work units (IGT_DEV_FLAGS=256 + IGT_DEV_TRACE).
Prints when the queues ran dry, how long the units took by class, and how many waves were busy over time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
path = '/tmp/igt_trace.bin'
os.environ['IGT_DEV_FLAGS'] = str(256 | int(os.environ.get('IGT_DEV_FLAGS', '0')))
os.environ['IGT_DEV_TRACE'] = path
import torch
from igtmpc import BatchSolver
from igtmpc.cinf import cinf_halfplanes
from igtmpc.scenarios import make_batch
DT = os.environ.get('IGT_PROBE_DTYPE', 'f64')
b = make_batch(B, dtype=np.float32 if DT == 'f32' else np.float64)
args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
with BatchSolver(dtype=DT, cand_mode=os.environ.get('IGT_PROBE_CAND', 'lattice')) as s:
    s.set_cinf(*cinf_halfplanes())           # the benchmark's configuration (bench.py): terminal set on
    for _ in range(3):
        s.solve(*args)
    torch.cuda.synchronize()
tr = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
if os.environ.get('IGT_TRACE_SAVE'):      # raw (scenario, slice, start, end) per unit, for offline analysis of the queue order
    W = 4 if DT == 'f64' else 2
    stride = ((B + 7) // 8) * W
    row = np.arange(len(tr)); ok = tr[:, 1] > 0
    qq = row // stride
    jj = (tr[:, 3] >> 8).astype(np.int64)
    bb = 8 * jj + ((qq - jj) & 7)
    np.savez_compressed(os.environ['IGT_TRACE_SAVE'], b=bb[ok], p=(tr[ok, 3] & 255).astype(np.int64), t0=tr[ok, 0], t1=tr[ok, 1],
                        wave=tr[ok, 2])
tr = tr[tr[:, 1] > 0]
t0 = tr[:, 0].astype(np.float64); t1 = tr[:, 1].astype(np.float64)
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0      # 100 MHz -> us
p = (tr[:, 3] & 255).astype(int)
dur = t1 - t0
print(f'B={B}: {len(tr)} units on {len(np.unique(tr[:, 2]))} waves; kernel span {t1.max():.1f} us; last unit START at {t0.max():.1f} us')
for pp in np.unique(p):
    d = dur[p == pp]
    print(f'  slice {pp}: n={len(d)} duration us: mean {d.mean():.1f} p50 {np.median(d):.1f} p90 {np.quantile(d, .9):.1f} p99 {np.quantile(d, .99):.1f} max {d.max():.1f}')
grid = np.linspace(0, t1.max(), 21)
busy = [(int(((t0 <= g) & (t1 > g)).sum())) for g in grid]
print('  busy waves at 5% steps of the span:', busy)
q = (tr[:, 2] % 8).astype(int)
for qq in range(8):
    m = q == qq
    print(f'  queue {qq}: units {m.sum()}  first start {t0[m].min():.1f}  last start {t0[m].max():.1f}  last end {t1[m].max():.1f}  sum of durations {dur[m].sum()/1e3:.2f} ms  mean dur slice0 {dur[m & (p == 0)].mean():.1f} slice1 {dur[m & (p == 1)].mean():.1f}')
# how good is the longest-first order?  duration vs position in the queue
order = np.argsort(t0)
k = len(order) // 10
print('  mean unit duration by start-time decile:', [round(float(dur[order[i * k:(i + 1) * k]].mean()), 1) for i in range(10)])
print('  p95 unit duration by start-time decile: ', [round(float(np.quantile(dur[order[i * k:(i + 1) * k]], .95)), 1) for i in range(10)])
late = np.argsort(-t1)[:8]
print('  last finishers: ' + ', '.join(f'(start {t0[i]:.0f} dur {dur[i]:.0f} slice {p[i]})' for i in late))
