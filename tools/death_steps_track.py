"""Analysis (oracle side, CPU): the tracking family's candidates -- at which control step does each fail its first verdict,
and how many wave-steps does whole-wave early exit execute when the 64-candidate units are cut along the steering-offset
axis (the lattice's cut) or along the acceleration-offset axis?       python tools/death_steps_track.py [B=256]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import np_oracle as O
from igtmpc.cinf import cinf_halfplanes
from igtmpc.scenarios import make_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, C, G = 20, 256, 16
P = O.Params(N=N)
sc = make_batch(B, N, P.dt, dtype=np.float64)
A, b = cinf_halfplanes(dt=P.dt, jerk=P.jerk)
for fam in ('track', 'lattice'):
    if fam == 'track':
        r = O.solve_batch_refined(sc['x0'], sc['u_prev'], sc['kparams'], sc['flags'], sc['obs_xy'], A, b, P, cand='track')[0]
    else:
        r = O.solve_batch(sc['x0'], sc['u_prev'], sc['kparams'], sc['flags'], sc['obs_xy'], A, b, P, return_all=True)
    X, U = r['X'], r['U']
    tol = P.feas_tol
    dead = np.full((B, C), N + 1, dtype=np.int64)      # step at whose bookkeeping the kernel knows (N+1: never)

    def mark(v):
        first = np.where(v.any(-1), v.argmax(-1), N + 1)
        np.minimum(dead, first, out=dead)
    v = X[..., O.IV, :N]
    mark(np.maximum(P.v_min - v, v - P.v_max) > tol)
    mark((np.abs(X[..., O.IEY, :N + 1]) - P.ey_lim) > tol)
    ob = sc['obs_xy']
    dx = X[:, :, None, O.IX, :] - ob[:, None, :, 0, :]; dy = X[:, :, None, O.IY, :] - ob[:, None, :, 1, :]
    col = (P.d_min ** 2 - (dx * dx + dy * dy)) > tol; col[..., 0] = False
    mark(col.any(2))
    t = A[:, 0] * X[..., O.IV, N - 1, None] + A[:, 1] * U[..., 0, N - 1, None] - b
    term = np.zeros((B, C, N + 1), bool); term[..., N - 1] = t.max(-1) > tol
    mark(term)
    c = np.arange(C); i, j = c // G, c % G
    def executed(groups):
        tot = 0.0
        for g in groups:
            last = dead[:, g].max(1)
            tot += np.minimum(last + 1, N).mean() / N
        return tot / len(groups)
    rank = np.empty(G, int); rank[np.argsort(np.abs(np.arange(G) - 7.5), kind='stable')] = np.arange(G)
    print(f'--- {fam}: feasible share of candidates {(dead > N).mean():.3f}, scenarios answered {(dead > N).any(1).mean():.3f}')
    print('alive by step:', np.round([(dead > k).mean() for k in range(0, N, 2)], 2))
    print('4 units by steering column, centre outwards (current): executed', round(executed([np.where(rank[j] // 4 == p)[0] for p in range(4)]), 3))
    print('4 units by acceleration offset, contiguous:            executed', round(executed([np.where(i // 4 == p)[0] for p in range(4)]), 3))
    print('4 units by acceleration offset, centre outwards:       executed', round(executed([np.where(rank[i] // 4 == p)[0] for p in range(4)]), 3))
    print('4 units 2x2 blocks (8 accel x 8 steer):                executed', round(executed([np.where((i // 8) * 2 + (j // 8) == p)[0] for p in range(4)]), 3))
    print('ideal (each candidate stops when it dies):             executed', round(np.minimum(dead + 1, N).mean() / N, 3))
    for ks in ([8], [6, 11], [4, 8, 12]):
        waves = np.full(B, 4.0); ex = 0.0; prev = 0
        for k in ks + [N]:
            ex += (waves * (k - prev)).mean(); prev = k
            if k < N: waves = np.ceil((dead > k).sum(1) / 64)
        print(f'scenario-wide compaction at {ks}: executed', round(ex / (4 * N), 3))
