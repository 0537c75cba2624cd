import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('igt-mpc-int_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import np_oracle as O
import igtmpc
from igtmpc.scenarios import make_batch
from igtmpc.cinf import cinf_halfplanes
from helpers import oracle_params, ambiguous_mask
B = 192
b = make_batch(B, dtype=np.float32)
f = lambda k: np.asarray(b[k], dtype=np.float64)
prev = O.candidates_lattice(f('u_prev'), O.Params())[np.arange(B), (np.arange(B) * 37) % 256]
u_ws = np.ascontiguousarray(O.shift_controls(prev).astype(np.float32))
u_prev = np.ascontiguousarray(prev[:, :, 0].astype(np.float32))
flags = b['flags'] | np.where(np.arange(B) % 3 != 0, 2, 0).astype(np.uint32)
args = (b['x0'], u_prev, b['kparams'], flags, b['obs_xy'])
outs = {}
for flag in ('0', '8388608', '262144'):
    os.environ['IGT_DEV_FLAGS'] = flag
    with igtmpc.BatchSolver(dtype='f32', cand_mode='track') as s:
        P = oracle_params(s)
        s.set_cinf(*cinf_halfplanes())
        outs[flag] = s.solve(*args, u_ws=u_ws)
        if flag == '0':
            allc = s.rollout_all(*args, u_ws=u_ws)
r0 = O.solve_batch_refined(f('x0'), u_prev.astype(np.float64), f('kparams'), flags, f('obs_xy'), *cinf_halfplanes(), P, u_ws=u_ws.astype(np.float64), cand='track')[0]
for flag, o in outs.items():
    bad = np.where(o['argmin'] != r0['argmin'])[0]
    print('flag', flag, 'mismatches vs oracle', len(bad), bad[:20])
    for i in bad[:8]:
        cg, co = o['argmin'][i], r0['argmin'][i]
        print('   scen', i, 'dev', cg, 'J', o['cost'][i], 'oracle', co, 'J', r0['cost'][i], '| oracle J of dev winner', r0['J'][i, cg] if cg >= 0 else None, 'feas', r0['feas'][i, cg] if cg >= 0 else None,
              '| device rollout_all cost of oracle winner', allc['cost'][i, co] if co >= 0 else None, 'viol', allc['viol'][i, co] if co >= 0 else None)
print('0 vs 8388608 equal:', all(np.array_equal(outs['0'][k], outs['8388608'][k], equal_nan=True) for k in ('x', 'u', 'cost', 'argmin', 'status')))
print('0 vs 262144 equal:', all(np.array_equal(outs['0'][k], outs['262144'][k], equal_nan=True) for k in ('x', 'u', 'cost', 'argmin', 'status')))
