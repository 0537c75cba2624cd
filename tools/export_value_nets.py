#!/usr/bin/env python3
"""Writes igt-mpc-int_amd/igtmpc/data/value_nets.npz: the weights of the eight terminal value networks the reference
ships (game_theoretic_NN/models/V_GT_sc{1..8}.pt, selected per scenario by sc{n}_config.yaml:2 `model_path`), as plain
arrays sc{n}_W{i} / sc{n}_b{i}.  The checkpoints are state dicts and are read with torch.load(weights_only=True): nothing
from the files is executed and no reference code is imported.  Runs in the build container only (/root/reference)."""
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for sc in range(1, 9):
    sd = torch.load(f'{REF}/game_theoretic_NN/models/V_GT_sc{sc}.pt', map_location='cpu', weights_only=True)
    keys = sorted([k for k in sd if k.endswith('weight')], key=lambda s: int(s.split('.')[1]))
    for li, k in enumerate(keys):
        out[f'sc{sc}_W{li}'] = sd[k].numpy()
        out[f'sc{sc}_b{li}'] = sd[k.replace('weight', 'bias')].numpy()
    print(f'sc{sc}: {len(keys) - 1} hidden layers, dtype {sd[keys[0]].dtype}, shapes {[tuple(sd[k].shape) for k in keys]}')
dst = os.path.join(ROOT, 'igt-mpc-int_amd', 'igtmpc', 'data', 'value_nets.npz')
np.savez_compressed(dst, **out)
print('written', dst, os.path.getsize(dst), 'bytes')
