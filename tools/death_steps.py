"""Analysis (oracle side, CPU): at which control step does each lattice candidate fail its first verdict, and how
many roll-out steps would different candidate->wave mappings save with whole-wave early exit or with
in-workgroup survivor compaction?  Informs the search kernel's work decomposition."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
from oracle import np_oracle as O
from igtmpc.scenarios import make_batch
from igtmpc.cinf import cinf_halfplanes

def death_steps(B=256, N=20, C=256, seed=0):
    P = O.Params(N=N)
    sc = make_batch(B, N, P.dt, seed=seed, dtype=np.float64)
    A, b = cinf_halfplanes(dt=P.dt, jerk=P.jerk)
    x0 = O.apply_flags(sc['x0'], sc['flags'])
    U = O.candidates_lattice(sc['u_prev'], P, C)                 # [B,C,2,N]
    X = O.rollout_frenet(x0[:, None, :], U, sc['kparams'][:, None, :], P)   # [B,C,7,N+1]
    tol = P.feas_tol
    dead = np.full((B, C), N + 1, dtype=np.int64)                # step at which the kernel would know (N+1 = never)
    def mark(viol_k):  # viol_k [B,C,K] for k = 0..K-1 -> step index where bit is first seen
        K = viol_k.shape[-1]
        first = np.where(viol_k.any(-1), viol_k.argmax(-1), N + 1)
        np.minimum(dead, first, out=dead)
    v = X[..., O.IV, :N]; mark(np.maximum(P.v_min - v, v - P.v_max) > tol)
    mark((np.abs(X[..., O.IEY, :N]) - P.ey_lim) > tol)
    ob = sc['obs_xy']                                            # [B,n_obs,2,N+1]
    dx = X[:, :, None, O.IX, :] - ob[:, None, :, 0, :]; dy = X[:, :, None, O.IY, :] - ob[:, None, :, 1, :]
    col = (P.d_min ** 2 - (dx * dx + dy * dy)) > tol; col[..., 0] = False
    mark(col.any(2)[..., :N])
    return dead, P

def saved(dead, groups, N):
    """groups: list of index arrays (one per wave).  Fraction of wave-steps executed with whole-wave early exit."""
    tot = 0.0
    for g in groups:
        last = dead[:, g].max(1)                                 # wave stops at the step where its last lane dies
        tot += np.minimum(last + 1, N).mean() / N
    return tot / len(groups)

if __name__ == '__main__':
    dead, P = death_steps()
    N = P.N; G = 16
    c = np.arange(256); i, j = c // G, c % G
    print('alive fraction by step:', np.round([(dead > k).mean() for k in range(N)], 2))
    cur = [np.where((c // 128) == p)[0] for p in range(2)]
    print('2 waves, current (accel halves):        executed', round(saved(dead, cur, N), 3))
    order = np.argsort(np.abs(j - 7.5), kind='stable')
    rank = np.empty(G, int); rank[np.argsort(np.abs(np.arange(G) - 7.5), kind='stable')] = np.arange(G)
    for nw in (2, 4):
        gs = [np.where(rank[j] // (G // nw) == p)[0] for p in range(nw)]
        print(f'{nw} waves by |steer increment| rank:       executed', round(saved(dead, gs, N), 3))
        gs = [np.where(j // (G // nw) == p)[0] for p in range(nw)]
        print(f'{nw} waves by steer increment (signed):    executed', round(saved(dead, gs, N), 3))
        gs = [np.where(i // (G // nw) == p)[0] for p in range(nw)]
        print(f'{nw} waves by accel increment:             executed', round(saved(dead, gs, N), 3))
    # in-workgroup compaction at steps k1,k2: waves needed = ceil(alive/64) (4 x 64 single) or ceil(alive/128) (2 x 128)
    for ks in ([8], [6, 11], [5, 9, 13], list(range(2, 20, 2))):
        for per in (64, 128):
            nw0 = 256 // per
            waves = np.full(dead.shape[0], nw0, float); ex = 0.0; prev = 0
            for k in ks + [N]:
                ex += (waves * (k - prev)).mean(); prev = k
                if k < N: waves = np.ceil((dead > k).sum(1) / per)
            print(f'compaction at {ks} ({per}/wave): executed', round(ex / (nw0 * N), 3))
