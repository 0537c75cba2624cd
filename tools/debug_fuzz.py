import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('igt-mpc-int_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import np_oracle as O
import igtmpc
from igtmpc.scenarios import make_batch
from helpers import oracle_params, rel_err
import test_gpu_fuzz as T
seed = int(sys.argv[1])
cfg = T._draw(seed)
print(cfg)
N, dt, C, B = cfg['N'], cfg['dt'], cfg['C'], cfg['B']
b = make_batch(max(B, 8), N=N, dt=dt, seed=100 + seed, dtype=np.float64)
b = {k: np.ascontiguousarray(v[:B]) for k, v in b.items() if isinstance(v, np.ndarray) and len(v) >= B}
obs = b['obs_xy']
more = [obs]
for m in range(1, cfg['n_obs']):
    lag = obs.copy(); lag[:, 0, 0, :] -= 9.0 * m * np.cos(0.3 * np.arange(B))[:, None]; lag[:, 0, 1, :] -= 9.0 * m * np.sin(0.3 * np.arange(B))[:, None]
    more.append(lag)
obs = np.ascontiguousarray(np.concatenate(more, axis=1))
with igtmpc.BatchSolver(N=N, dt=dt, n_rk4=cfg['n_rk4'], C=C, n_obs=cfg['n_obs'], dtype='f64', cand_mode=cfg['cand'], **cfg['limits']) as s:
    P = oracle_params(s)
    n_all = min(B, 4)
    allc = s.rollout_all(b['x0'][:n_all], b['u_prev'][:n_all], b['kparams'][:n_all], b['flags'][:n_all], obs[:n_all])
first = O.solve_batch_refined(b['x0'], b['u_prev'], b['kparams'], b['flags'], obs, None, None, P, C=C, refine_iters=0, cand=cfg['cand'])[0]
X = first['X'][:n_all]
err = rel_err(allc['X'], X)                       # [n,C,7,N+1]
print('U max err', rel_err(allc['U'], first['U'][:n_all]).max())
e = err.max(axis=(2, 3))
for bi in range(n_all):
    bad = np.where(e[bi] > 1e-9)[0]
    print('scenario', bi, 'kp', b['kparams'][bi], 'bad candidates', len(bad))
    for c in bad[:6]:
        k_first = np.argmax(err[bi, c].max(0) > 1e-9)
        print('  c', c, 'max err', e[bi, c], 'first bad step', k_first, 'ey range', X[bi, c, 3].min(), X[bi, c, 3].max(), 'epsi range', X[bi, c, 4].min(), X[bi, c, 4].max(),
              's at bad', X[bi, c, 2, max(k_first - 1, 0):k_first + 1], 'v', X[bi, c, 5, max(k_first - 1, 0)], 'feas', first['feas'][bi, c])
        print('     err by row at first bad:', err[bi, c, :, k_first])
