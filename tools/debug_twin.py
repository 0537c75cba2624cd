#!/usr/bin/env python3
"""debug: tracking N=40 B=1500 with / without live-row masks (IGT_DEV_FLAGS 0 / 2097152 / 8388608), refine 0 / 1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'igt-mpc-int_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import np_oracle as O
import igtmpc as igt
from igtmpc.scenarios import make_batch
from igtmpc.cinf import cinf_halfplanes
N, C, B = 40, 256, 1500
b = make_batch(B, dtype=np.float64, N=N)
prev = O.candidates_lattice(b['u_prev'], O.Params(N=N))[np.arange(B), (np.arange(B) * 37) % 256]
u_ws = np.ascontiguousarray(O.shift_controls(prev)); u_prev = np.ascontiguousarray(prev[:, :, 0])
flags = b['flags'] | np.where(np.arange(B) % 3 != 0, 2, 0).astype(np.uint32)
for refine in (0, 1):
    outs = {}
    for flag in ('0', '2097152', '8388608'):
        os.environ['IGT_DEV_FLAGS'] = flag
        with igt.BatchSolver(N=N, C=C, dtype='f64', cand_mode='track', refine_iters=refine) as s:
            s.set_cinf(*cinf_halfplanes())
            outs[flag] = s.solve(b['x0'], u_prev, b['kparams'], flags, b['obs_xy'], u_ws=u_ws)
    for flag in ('2097152', '8388608'):
        o0, o = outs['0'], outs[flag]
        da = o0['argmin'] != o['argmin']
        dc = ~np.isclose(o0['cost'], o['cost'], rtol=0, atol=0, equal_nan=True)
        dx = ~np.all((o0['x'] == o['x']) | (np.isnan(o0['x']) & np.isnan(o['x'])), axis=(1, 2))
        print(f'refine {refine} flags 0 vs {flag}: argmin differs {da.sum()}  cost differs {dc.sum()}  x differs {dx.sum()}')
        for i in np.nonzero(da | dc | dx)[0][:6]:
            print('   b', i, 'argmin', o0['argmin'][i], o['argmin'][i], 'rows', o0['argmin'][i] // 16, o['argmin'][i] // 16, 'cost', o0['cost'][i], o['cost'][i],
                  'max|dx|', np.nanmax(np.abs(o0['x'][i] - o['x'][i])), 'max|du|', np.nanmax(np.abs(o0['u'][i] - o['u'][i])))
