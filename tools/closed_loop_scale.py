#!/usr/bin/env python3
"""Developer probe (GPU box): device-resident closed loop with E episodes in lock-step (tracking candidates, warm start,
f64, N = 20, 50 steps), eager and replayed from a stream graph.      python tools/closed_loop_scale.py [episodes ...]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
from igtmpc.evaluate import run_closed_loop
steps = 50
for E in ([int(a) for a in sys.argv[1:]] or [16, 256, 2048, 8192]):
    row = {'episodes': E, 'problems_per_step': 2 * E}
    for name, g in (('eager', False), ('graph', True)):
        run_closed_loop(sc=1, num_samples=E, N=20, T_sim=steps / 10, device_resident=True, graph=g)      # warm-up (first-touch costs)
        r = run_closed_loop(sc=1, num_samples=E, N=20, T_sim=steps / 10, device_resident=True, graph=g)
        row[f'{name}_ms_per_step'] = round(r['wall_s'] / steps * 1e3, 3)
        row[f'{name}_agent_steps_per_s'] = round(2 * E * steps / r['wall_s'])
    row['infeasible'] = float(r['infeasible_ratio'].mean().round(3))
    print(json.dumps(row), flush=True)
