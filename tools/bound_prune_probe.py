"""Analysis (oracle side, CPU): how many wave-steps of the search would a cost bound save?
A candidate's final cost is  J = sum of non-negative stage terms - (s_N - s_0)  (mpc.py:361-364, 372); the progress still to come
after step k is bounded by the row's own speed profile (ds/dt <= |v| / (1 - |K| |ey|), |ey| <= 0.3 at any stage of a state that can
still be feasible), so  LB_k = (stage terms up to k) - (s_k - s_0) - 1.04 sum_{k' >= k} dt max(|v_k'|, |v_k'+1|)  is a lower bound of J.
A candidate with LB_k > J_inc (the cost of a feasible candidate of the same scenario already rolled) cannot win.
    python tools/bound_prune_probe.py [B=256]"""
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
SLACK = 1.04
for fam in ('lattice', 'track'):
    if fam == 'track':
        r = O.solve_batch_refined(sc['x0'], sc['u_prev'], sc['kparams'], sc['flags'], sc['obs_xy'], A, b, P, cand='track')[0]
    else:
        r = O.solve_batch(sc['x0'], sc['u_prev'], sc['kparams'], sc['flags'], sc['obs_xy'], A, b, P, return_all=True)
    X, U = r['X'], r['U']
    tol = P.feas_tol
    dead = np.full((B, C), N + 1, dtype=np.int64)

    def mark(v, into):
        first = np.where(v.any(-1), v.argmax(-1), N + 1)
        np.minimum(into, first, out=into)
    v = X[..., O.IV, :]
    mark(np.maximum(P.v_min - v[..., :N], v[..., :N] - P.v_max) > tol, dead)
    mark((np.abs(X[..., O.IEY, :N + 1]) - P.ey_lim) > tol, dead)
    ob = sc['obs_xy']
    dx = X[:, :, None, O.IX, :] - ob[:, None, :, 0, :]; dy = X[:, :, None, O.IY, :] - ob[:, None, :, 1, :]
    col = (P.d_min ** 2 - (dx * dx + dy * dy)) > tol; col[..., 0] = False
    mark(col.any(2), dead)
    t = A[:, 0] * X[..., O.IV, N - 1, None] + A[:, 1] * U[..., 0, N - 1, None] - b
    term = np.zeros((B, C, N + 1), bool); term[..., N - 1] = t.max(-1) > tol
    mark(term, dead)
    feas = dead > N
    # live rows (accel_rows_kernel): speed box over k < N and the terminal set -- a function of the row alone
    c = np.arange(C); i, j = c // G, c % G
    row_dead = ((np.maximum(P.v_min - v[..., :N], v[..., :N] - P.v_max) > tol).any(-1) | term[..., N - 1])     # [B,C]
    live_row = ~row_dead.reshape(B, G, G).all(2)                                                              # [B,G]
    # costs: stage terms in the kernel's order; J_k = what is on the books when step k's bookkeeping is done (k = 0..N)
    ep, ey, s = X[..., O.IEPSI, :], X[..., O.IEY, :], X[..., O.IS, :]
    stage = ep ** 2 + ey ** 2
    stage[..., :N] += P.w_u * (U[..., 0, :] ** 2 + U[..., 1, :] ** 2)
    Jk = np.cumsum(stage, -1)                                                       # [B,C,N+1]
    J = Jk[..., N] - (s[..., N] - s[..., 0])
    assert np.allclose(np.where(feas, J, 0), np.where(feas, r['J'], 0), atol=1e-9)
    Jbest = np.where(feas, J, np.inf).min(1)                                        # [B]
    inc = np.abs(v)
    seg = P.dt * np.maximum(inc[..., :N], inc[..., 1:])                             # bound of s_{k+1} - s_k
    rem = SLACK * np.concatenate([np.cumsum(seg[..., ::-1], -1)[..., ::-1], np.zeros((B, C, 1))], -1)   # [B,C,N+1]: progress still to come at k
    LB = Jk - (s - s[..., :1]) - rem
    ok = (s[..., 1:] - s[..., :-1] <= SLACK * seg + 1e-12) | ~feas[..., None]
    assert ok.all(), 'the progress bound does not hold for a feasible candidate'
    assert (LB[..., N][feas] <= J[feas] + 1e-9).all()

    def bound_dead(Jinc):
        d = np.full((B, C), N + 1, dtype=np.int64)
        mark(LB > Jinc[:, None, None], d)
        return d
    rank = np.empty(G, int); rank[np.argsort(np.abs(np.arange(G) - 7.5), kind='stable')] = np.arange(G)

    def units_of(bi):
        """the device's units of scenario bi: lists of candidate indices (igt_kernels_common.h unit_layout / unit_candidate)"""
        rows = np.where(live_row[bi])[0]
        R = len(rows)
        if R == 0:
            return []
        if fam == 'track':
            return [np.array([r_ * G + jj for r_ in rows[p * 4:(p + 1) * 4] for jj in range(G)]) for p in range((R + 3) // 4)]
        cols = [G // 2 - 1 - (r_ >> 1) if (r_ & 1) else G // 2 + (r_ >> 1) for r_ in range(G)]
        allc = np.array([row * G + col for col in cols for row in rows])
        return [allc[p * 64:(p + 1) * 64] for p in range((len(allc) + 63) // 64)]

    def executed(dead_by, first_unit_blind):
        """wave-steps executed / (4 N) per scenario; first_unit_blind: unit 0 of a scenario rolls without an incumbent"""
        tot = 0.0
        for bi in range(B):
            for p, g in enumerate(units_of(bi)):
                d = dead[bi, g] if (first_unit_blind and p == 0) else np.minimum(dead[bi, g], dead_by[bi, g])
                tot += min(d.max() + 1, N)
        return tot / (B * 4 * N)
    none = np.full((B, C), N + 1)
    print(f'--- {fam}: feasible share {feas.mean():.3f}; live rows {live_row.sum(1).mean():.1f}; units per scenario '
          f'{np.mean([len(units_of(bi)) for bi in range(B)]):.2f}')
    print('today (verdicts only):                                       executed', round(executed(none, False), 3))
    dB = bound_dead(Jbest)
    print('bound against the final best cost, every unit (upper limit): executed', round(executed(dB, False), 3))
    print('bound against the final best, unit 0 rolls blind:            executed', round(executed(dB, True), 3))
    # incumbent = best of unit 0 only (what the other units would see if unit 0 has finished)
    J0 = np.full(B, np.inf)
    for bi in range(B):
        u = units_of(bi)
        if u:
            J0[bi] = np.where(feas[bi, u[0]], J[bi, u[0]], np.inf).min()
    print('bound against unit 0\'s best, unit 0 blind:                   executed', round(executed(bound_dead(J0), True), 3))
    print('   share of scenarios whose winner is in unit 0:', round(np.mean(J0[np.isfinite(Jbest)] == Jbest[np.isfinite(Jbest)]), 3))
    # a-priori: rows that lose before a step is rolled (LB_0 > incumbent)
    print('   candidates of live rows dead by the bound at k = 0 (final best):', round(((dB == 0) & np.repeat(live_row, G, 1)).sum() / np.repeat(live_row, G, 1).sum(), 3))
    if fam == 'track':
        # the scheme that can be built: units in order of DEscending row rank (the highest live rows first); the first rolls blind, each
        # later one sees the best cost of those before it; slack 1 on scenarios that cannot meet their arc, 1 / (1 - |kv| (ey_lim + dt vabs)) else
        kp = sc['kparams']
        vabs = max(abs(P.v_min), abs(P.v_max)) + max(abs(P.a_min), abs(P.a_max)) * P.dt
        reach = N * P.dt * vabs * 1.2
        s0 = sc['x0'][:, 2]
        clear = (kp[:, 2] == 0) | (s0 + reach < kp[:, 0]) | (s0 - reach > kp[:, 1])
        slack = np.where(clear, 1.0, 1.0 / (1.0 - np.abs(kp[:, 2]) * (P.ey_lim + P.dt * vabs)))
        rem2 = slack[:, None, None] * np.concatenate([np.cumsum(seg[..., ::-1], -1)[..., ::-1], np.zeros((B, C, 1))], -1)
        LB2 = Jk - (s - s[..., :1]) - rem2
        tot = 0.0; tot_unit = 0
        for bi in range(B):
            us = units_of(bi)[::-1]
            Jinc = np.inf
            for g in us:
                d = np.full(len(g), N + 1)
                hit = LB2[bi, g] > Jinc
                d = np.where(hit.any(-1), hit.argmax(-1), N + 1)
                d = np.minimum(d, dead[bi, g])
                tot += min(d.max() + 1, N)
                Jinc = min(Jinc, np.where(feas[bi, g], J[bi, g], np.inf).min())
        print(f'top rows first, each unit sees the units before it (clear share {clear.mean():.2f}): executed', round(tot / (B * 4 * N), 3))
        top = [np.where(feas[bi, units_of(bi)[-1]], J[bi, units_of(bi)[-1]], np.inf).min() == Jbest[bi] for bi in range(B)
               if np.isfinite(Jbest[bi]) and units_of(bi)]
        print('   share of scenarios whose winner is in the unit of the highest live rows:', round(float(np.mean(top)), 3))
