#!/usr/bin/env python3
"""The benchmark scenarios NO candidate family answers (20 of 256 in profiles/r02_nlp_gap.txt): infeasible problems, or
feasible ones the families miss?  CPU only.  For each such scenario
  1. a certificate of infeasibility that needs no solver: every state a speed-, rate- and lane-feasible plan can be in at
     step k lies on the route between s_lo(k) and s_hi(k) (slowest / fastest jerk-limited speed profile from (v_0, a_prev)),
     within |e_y| <= 0.2 of the centre line (mpc.py:296-299) -- if the forecast circle of radius d_min (mpc.py:223-226)
     covers that whole strip at some k, every plan collides there;
  2. otherwise the reference's NLP itself (oracle/nlp_quality.py: scipy SLSQP on the single-shooting restatement of
     mpc.py:147-160) from cold starts -- u_prev held, the jerk-limited brake and acceleration ramps, and the least-violating
     candidate of the tracking family -- keeping the smallest constraint violation any start reaches.
    python tools/nlp_unanswered.py [n_scenarios=256]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import np_oracle as O
import nlp_quality as Q
from igtmpc import routes as R
from igtmpc.cinf import cinf_halfplanes
from igtmpc.scenarios import make_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = O.Params()
cinf = cinf_halfplanes()
b = {k: (np.asarray(v, dtype=np.float64) if v.dtype.kind == 'f' else v) for k, v in make_batch(n, dtype=np.float64).items()}
args = (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'], *cinf, P)
fams = [O.solve_batch(*args), O.solve_batch_refined(*args, refine_iters=2)[-1], O.solve_batch_refined(*args, refine_iters=2, cand='track')[-1]]
answered = np.zeros(n, bool)
for f in fams:
    answered |= f['status'] == 0
todo = np.nonzero(~answered)[0]
print(f'{n} scenarios, {answered.sum()} answered by at least one family, {len(todo)} by none')
track = O.solve_batch_refined(*args, cand='track')[0]           # every candidate's worst violation g, for a warm-ish start


def route_of(i):
    """route id of scenario i from its curvature parameters and start pose (the generator does not keep it)."""
    best, arg = np.inf, -1
    for rid in range(12):
        kp = R.kparams(rid)
        same = np.array_equal(np.nan_to_num(kp, posinf=1e9), np.nan_to_num(b['kparams'][i], posinf=1e9))
        if not same:
            continue
        p = R.frenet2global(rid, b['x0'][i, 2])
        d = np.hypot(p[0] - b['x0'][i, 0], p[1] - b['x0'][i, 1])
        if d < best:
            best, arg = d, rid
    return arg, best


def strip_certificate(i):
    rid, off = route_of(i)
    if rid < 0 or off > 0.25:
        return None
    s0, v0, a_prev = b['x0'][i, 2], b['x0'][i, 5], b['u_prev'][i, 0]
    ra = P.dt * P.jerk
    lo = hi = s0
    vlo = vhi = v0
    for k in range(1, P.N + 1):
        alo = max(P.a_min, a_prev - ra * k)
        ahi = min(P.a_max, a_prev + ra * k)
        lo += max(0.0, min(vlo, vlo + alo * P.dt)) * P.dt       # never faster than the slow profile's slower end
        hi += min(P.v_max + P.a_max * P.dt, max(vhi, vhi + ahi * P.dt)) * P.dt
        vlo = max(P.v_min, vlo + alo * P.dt)
        vhi = min(P.v_max, vhi + ahi * P.dt)
        # ds/dt = v cos(beta + epsi) / (1 - K e_y): within [0.90, 1.05] v for a plan that stays in the lane (|K e_y| <= 0.024,
        # a heading error beyond 0.4 rad leaves the lane within a step or two) -- the strip is widened accordingly
        slo, shi = s0 + 0.90 * (lo - s0) - 0.05, s0 + 1.05 * (hi - s0) + 0.05
        ss = np.linspace(slo, shi, 64)
        pts = R.frenet2global(np.full(64, rid), ss)
        for o in b['obs_xy'][i]:
            d = np.hypot(pts[:, 0] - o[0, k], pts[:, 1] - o[1, k])
            # the sampled centre-line points are at most (hi - lo + 0.1) / 63 / 2 apart from any point of the strip's axis
            slack = P.ey_lim + (shi - slo) / 63 / 2 + off
            if (d + slack < P.d_min).all():
                return k
    return None


def cold_starts(i):
    N, ra, rd = P.N, P.dt * P.jerk, P.dt * P.steer_rate
    a0, d0 = b['u_prev'][i]
    k = np.arange(1, N + 1)
    hold = np.stack([np.full(N, a0), np.full(N, d0)])
    brake = np.stack([np.maximum(a0 - ra * k, P.a_min), np.full(N, d0)])
    # brake, then come back to a = 0 as the speed runs out (v >= 0 must hold, mpc.py:316)
    soft = brake.copy()
    v = b['x0'][i, 5]
    for j in range(N):
        if v + soft[0, j] * P.dt < 0.3:
            soft[0, j:] = np.minimum(0.0, soft[0, j - 1] + ra * np.arange(1, N - j + 1)) if j > 0 else 0.0
            break
        v += soft[0, j] * P.dt
    accel = np.stack([np.minimum(a0 + ra * k, P.a_max), np.full(N, d0)])
    c = int(np.argmin(np.where(np.isfinite(track['g'][i]), track['g'][i], np.inf)))
    return [hold, brake, soft, accel, track['U'][i, c]]


n_cert, n_feasible, n_open, rows = 0, 0, 0, []
for i in todo:
    k = strip_certificate(i)
    if k is not None:
        n_cert += 1
        rows.append((i, f'infeasible: the forecast covers every reachable state at step {k}'))
        continue
    best = np.inf
    for u0 in cold_starts(i):
        r = Q.polish(b['x0'][i], b['u_prev'][i], b['kparams'][i], b['flags'][i], b['obs_xy'][i], cinf[0], cinf[1], P, u0, maxiter=150)
        best = min(best, r['max_violation'])
        if best < 1e-6:
            break
    if best < 1e-6:
        n_feasible += 1
        rows.append((i, f'FEASIBLE for the NLP (violation {best:.1e}): missed by the families'))
    else:
        n_open += 1
        g = track['g'][i]
        rows.append((i, f'no feasible point found from 5 cold starts (smallest violation {best:.3f}; best candidate of the tracking '
                        f'family violates by {np.nanmin(g):.3f}; verdict bits of that candidate {int(track["mask"][i, int(np.nanargmin(g))]):#04x})'))
for i, msg in rows:
    print(f'scenario {i:4d}: {msg}')
print(f'{len(todo)} unanswered: {n_cert} certified infeasible, {n_feasible} feasible for the NLP (missed), {n_open} undecided '
      f'(no certificate, no feasible point from cold starts)')
