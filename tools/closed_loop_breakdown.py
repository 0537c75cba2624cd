#!/usr/bin/env python3
"""Developer probe (GPU box): what the infeasible steps of the closed loop are made of.  An agent-step is infeasible when
the solve returned no candidate and the brake fallback was applied (evaluate.py:511-545).  Classes, in this order:
  stopped   the vehicle stands (v = 0) and stays infeasible: gridlock within d_min of the other vehicle, or stranded with
            |ey| > ey_lim (constraints that include the current state, mpc.py:223-226 / 296-299)
  cascade   follows an infeasible step while still braking (u_prev.a = -4: no candidate keeps v >= 0 over the horizon)
  first     the step that started a cascade
    python tools/closed_loop_breakdown.py [cand_mode] [N] [C] [refine_iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import numpy as np
from igtmpc.evaluate import run_closed_loop
EPISODES = int(os.environ.get('IGT_CL_EPISODES', '64'))        # episodes per scenario
cand = sys.argv[1] if len(sys.argv) > 1 else 'track'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
C = int(sys.argv[3]) if len(sys.argv) > 3 else 256
RI = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tot = dict(steps=0, infeasible=0, stopped=0, gridlock=0, stranded=0, cascade=0, first=0)
for sc in range(1, 9):
    r = run_closed_loop(sc=sc, num_samples=EPISODES, N=N, cand_mode=cand, dtype='f64', C=C, refine_iters=RI)
    x, u = r['x_data'], r['u_data']                 # [E, 14, T+1], [E, 4, T]
    E, _, T = u.shape
    for m in range(2):
        a = u[:, 2 * m, :]
        v = x[:, 7 * m + 5, :-1]
        ey = x[:, 7 * m + 3, :-1]
        d = np.hypot(x[:, 0, :-1] - x[:, 7, :-1], x[:, 1, :-1] - x[:, 8, :-1])
        # infeasible steps: the applied input is the fallback's (a = -4 while v > 0, or a = 0 with v = 0 after a stop)
        brake = np.isclose(a, -4.0) & (v > 0)
        # a stopped vehicle that remains infeasible applies a = 0 and does not move
        still = (np.abs(v) < 1e-12) & np.isclose(a, 0.0) & (np.abs(x[:, 7 * m + 5, 1:]) < 1e-12) & (np.arange(T)[None, :] > 0)
        prev_inf = np.zeros_like(brake)
        prev_inf[:, 1:] = (brake | still)[:, :-1]
        stopped = still & prev_inf
        # propagate "stopped" forward: a standing vehicle whose previous step was stopped
        for t in range(1, T):
            stopped[:, t] |= still[:, t] & stopped[:, t - 1]
        cascade = brake & prev_inf
        first = brake & ~prev_inf
        inf = brake | stopped
        tot['steps'] += E * T
        tot['infeasible'] += int(inf.sum()); tot['stopped'] += int(stopped.sum()); tot['cascade'] += int(cascade.sum())
        tot['first'] += int(first.sum())
        tot['gridlock'] += int((stopped & (d < 5.6)).sum()); tot['stranded'] += int((stopped & (np.abs(ey) > 0.2)).sum())
    chk = r['infeasible_ratio'].sum() * T
    print(f'sc {sc}: driver-counted infeasible agent-steps {chk:.0f}', flush=True)
print(f'cand={cand} N={N} C={C} refine={RI}: {tot}')
i = max(tot['infeasible'], 1)
print(f"share of infeasible agent-steps: first {tot['first'] / i:.3f}  cascade {tot['cascade'] / i:.3f}  stopped {tot['stopped'] / i:.3f} "
      f"(gridlock within d_min {tot['gridlock'] / i:.3f}, stranded outside the lane {tot['stranded'] / i:.3f});  "
      f"infeasible / all steps {tot['infeasible'] / tot['steps']:.3f};  first failures per 150-step agent run {tot['first'] / (tot['steps'] / 150):.2f}")
