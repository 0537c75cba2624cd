#!/usr/bin/env python3
"""Developer probe (GPU box): what the speed cap of the tracking family's targets (igt_params.track_vcap) costs and buys --
search time with the cap on / off, with and without the incumbent bound (IGT_DEV_FLAGS = 8388608), and the mean cost of the
answers.      python tools/speed_cap_probe.py [B=65536] [dtype=f64]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd'))
import torch
from igtmpc import BatchSolver
from igtmpc.cinf import cinf_halfplanes
from igtmpc.scenarios import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dtype = sys.argv[2] if len(sys.argv) > 2 else 'f64'
b = make_batch(B, dtype=np.float64 if dtype == 'f64' else np.float32)
args = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda() for a in (b['x0'], b['u_prev'], b['kparams'], b['flags'], b['obs_xy'])]
ref = None
for flag in ('0', '8388608'):
    os.environ['IGT_DEV_FLAGS'] = flag
    for vcap in (0.0, 1.0):
        with BatchSolver(dtype=dtype, cand_mode='track', track_vcap=vcap) as s:
            s.set_cinf(*cinf_halfplanes())
            s.set_profiling(True)
            out = s.solve(*args)
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                s.solve(*args, out=out); torch.cuda.synchronize()
                ts.append(s.kernel_ms()[0])
            ok = (out['status'] == 0)
            cost = out['cost'][ok].double().mean().item()
        print(f'{dtype} B={B} IGT_DEV_FLAGS={flag:8s} track_vcap={vcap:.0f}: search {np.median(ts):.3f} ms   solved {ok.float().mean().item():.3f}  mean cost {cost:.4f}', flush=True)
