"""The terminal value networks the reference ships, as data (game_theoretic_NN/models/V_GT_sc{1..8}.pt; one per
scenario, chosen by sc{n}_config.yaml:2 `model_path` -- mpc.py:108-124 loads it into `mlp(6, 1, [128]*num_layers, tanh)`).
data/value_nets.npz holds their weights as plain float64 arrays (tools/export_value_nets.py: state dicts read with
torch.load(weights_only=True)); the feature / target normalisation statistics (mpc.py:110-118) come from a dataset the
reference does not ship, so callers get the layers only and BatchSolver.set_value_net applies identity statistics unless
told otherwise."""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'value_nets.npz')
_CACHE = {}


def shipped_value_net(sc):
    """-> dict(layers=[(W[out,in], b[out]), ...]) of scenario sc (1..8): 2 hidden layers of 128 for sc 1, 2, 4, 5, 8 and
    3 for sc 3, 6, 7."""
    sc = int(sc)
    if not 1 <= sc <= 8:
        raise ValueError('scenario must be 1..8')
    if sc not in _CACHE:
        with np.load(_PATH) as z:
            layers, i = [], 0
            while f'sc{sc}_W{i}' in z:
                layers.append((np.array(z[f'sc{sc}_W{i}']), np.array(z[f'sc{sc}_b{i}'])))
                i += 1
        _CACHE[sc] = layers
    return dict(layers=[(W.copy(), b.copy()) for W, b in _CACHE[sc]])
