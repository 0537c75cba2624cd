#!/usr/bin/env python3
"""Which constraint empties the candidate set when a closed-loop solve fails?  CPU only (oracle/closed_loop.py, the
float64 restatement of evaluate.py:451-569 with the oracle's shooting solve): 8 scenarios x E episodes x 150 steps,
tracking candidates + warm start.  For every FIRST failure of an agent (the step before was solved) the verdict bits of
all 256 candidates are tallied, and the two profiles that bound what the jerk limit lets the vehicle reach within the
horizon (acceleration ramping up / down at the limit throughout, each with the 16 steering offsets) are tried on the
same problem: if they fail too, no acceleration profile avoids the forecast.       python tools/first_failure_verdicts.py [track_env=1.0] [episodes per scenario=4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import np_oracle as O
import closed_loop as CL
from igtmpc import evaluate as EV, routes as R
from igtmpc.cinf import cinf_halfplanes

ENV = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = 150
P, cinf = O.Params(), cinf_halfplanes()
BITS = dict(box_v=1, box_u=2, rate=4, ey=8, terminal=16, collision=32, nonfinite=64)
log = []
_solve = O.solve_batch_refined


def recording_solve(*a, **k):
    k['track'] = dict(env=ENV)
    res = _solve(*a, **k)
    log.append((int(res[-1]['status'][0]), res[-1]['mask'][0].copy(), a[:5]))
    return res


def boundary_profiles_feasible(x0, u_prev, kp, flags, obs):
    """Full-jerk ramp up and full-jerk ramp down (all 16 acceleration rows collapse to it), 16 steering offsets each."""
    for sign in (+1.0, -1.0):
        U = O.candidates_track(O.apply_flags(x0, flags), u_prev, kp, np.array([[sign * 100.0, 0.0]]), np.array([[0.0, 0.1]]), True, P)
        r = O.solve_batch(x0, u_prev, kp, flags, obs, *cinf, P, U=U, generated=True)
        if r['status'][0] == 0:
            return True
    return False


O.solve_batch_refined = recording_solve
tot = dict(agent_steps=0, infeasible=0, first=0, first_with_a_feasible_boundary_profile=0)
share = {k: 0.0 for k in BITS}
every = {k: 0 for k in BITS}
for sc in range(1, 9):
    rng = np.random.default_rng(2026)
    pairs = [R.SCENARIO_ROUTES[sc - 1][e % 4] for e in range(NE)]
    x, _ = EV.initial_states(rng, pairs)
    for e in range(NE):
        log.clear()
        CL.run_episode(x[e], pairs[e], P, cinf, M_sim=T, cand_mode='track')
        st = np.array([l[0] for l in log]).reshape(T, 2)            # the loop solves agent 0 then agent 1, step by step
        tot['agent_steps'] += 2 * T
        tot['infeasible'] += int((st != 0).sum())
        for i in range(2):
            for t in range(T):
                if st[t, i] != 0 and (t == 0 or st[t - 1, i] == 0):
                    tot['first'] += 1
                    m = log[2 * t + i][1]
                    tot['first_with_a_feasible_boundary_profile'] += int(boundary_profiles_feasible(*log[2 * t + i][2]))
                    for k, bit in BITS.items():
                        share[k] += float(((m & bit) != 0).mean())
                        every[k] += int(((m & bit) != 0).all())
    print(f'sc {sc}: {tot}', flush=True)
n = max(tot['first'], 1)
print(f"track_env = {ENV}: {tot}; infeasible agent-steps {tot['infeasible'] / tot['agent_steps']:.3f}, "
      f"first failures per 150-step agent run {tot['first'] / (tot['agent_steps'] / T):.2f}")
print('at a first failure, share of the 256 candidates carrying each verdict bit:', {k: round(v / n, 3) for k, v in share.items()})
print(f'first failures at which EVERY candidate carries the bit (of {tot["first"]}):', every)
