#!/usr/bin/env python3
"""When each of the K solves of a timed region completes, without a profiler: bench.py's lanes (F handles on F streams, the
same batch) driven through settle -> fence -> K steps -> fence with an event recorded behind every solve.

  usage: region_events.py [K=20] [F=4] [policy=drain|none|ramp|both] [reps=5] [noev]
policy: what igt_set_concurrency is told during the region --
  none   F throughout;   drain  min(F, steps left) (bench.py);   ramp  min(F, steps issued so far + 1);   both  the smaller
Prints, per repetition, the region's wall time and the completion time of every step; then the mean."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import numpy as np, torch
from igtmpc import BatchSolver
from igtmpc.cinf import cinf_halfplanes
from igtmpc.scenarios import make_batch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
policy = sys.argv[3] if len(sys.argv) > 3 else 'drain'
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
use_events = not (len(sys.argv) > 5 and sys.argv[5] == 'noev')      # 'noev': only the wall clock, nothing recorded on the lanes
B, N, C = 4096, 20, 256
batch = make_batch(B, N=N, dtype=np.float64)
dargs = [torch.from_numpy(batch[k].view(np.int32) if batch[k].dtype == np.uint32 else batch[k]).cuda() for k in ('x0', 'u_prev', 'kparams', 'flags', 'obs_xy')]
solvers, outs, lanes = [], [], []
for _ in range(F):
    sv = BatchSolver(N=N, C=C, n_obs=1, device=0, dtype='f64')
    sv.set_cinf(*cinf_halfplanes(dt=sv.params.dt, jerk=sv.params.jerk_limit))
    sv.set_concurrency(F)
    solvers.append(sv); outs.append(sv.solve(*dargs)); lanes.append(torch.cuda.Stream())
torch.cuda.synchronize()
conc = [F] * F
n = [0]
def step(want=None, ev=None):
    q = n[0] % F; n[0] += 1
    want = F if want is None else max(1, min(F, want))
    if want != conc[q]:
        solvers[q].set_concurrency(want); conc[q] = want
    with torch.cuda.stream(lanes[q]):
        solvers[q].solve(*dargs, out=outs[q])
        if ev is not None:
            ev.record()
def hint(i):
    d, r = K - i, i + 1
    return {'none': None, 'drain': d, 'ramp': r, 'both': min(d, r)}[policy]
for _ in range(800):
    step()
torch.cuda.synchronize()
walls, ends = [], []
for rep in range(reps):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for i in range(K):
        step(hint(i), evs[i] if use_events else None)
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    done = [e0.elapsed_time(e) for e in evs] if use_events else [0.0] * K
    walls.append(wall * 1e3); ends.append(done)
    print(f'rep {rep}: wall {wall * 1e3:.3f} ms  (issued in {issued * 1e3:.3f})  ends: ' + ' '.join(f'{d:.2f}' for d in done))
print(f'policy {policy} F={F} K={K}: mean wall {np.mean(walls):.3f} ms  min {np.min(walls):.3f}  -> {B * K / np.mean(walls) / 1e3:.2f} M solves/s;  '
      f'mean completion of step K-4..K-1: ' + ' '.join(f'{v:.2f}' for v in np.mean(np.array(ends), axis=0)[-4:]))
