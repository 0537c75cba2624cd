#!/usr/bin/env python3
"""Developer probe (GPU box): device-resident closed loop with thousands of episodes in lock-step."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
from igtmpc.evaluate import run_closed_loop
for E in (256, 2048, 8192):
    r = run_closed_loop(sc=1, num_samples=E, N=20, T_sim=5.0, device_resident=True)
    steps = 50
    print(json.dumps({'episodes': E, 'problems_per_step': 2 * E, 'ms_per_step': round(r['wall_s'] / steps * 1e3, 3),
                      'agent_steps_per_s': round(2 * E * steps / r['wall_s']), 'infeasible': r['infeasible_ratio'].mean().round(3)}), flush=True)
