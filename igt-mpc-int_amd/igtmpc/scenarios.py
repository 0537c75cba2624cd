"""Synthetic batched two-vehicle intersection scenarios.

Deterministic (np.random.default_rng(2026), the seed evaluate.py:35 uses).  For batch
index b: combo = b mod 64 -> scenario sc = combo//8 + 1, rotation (combo//2) mod 4 picks
one of the scenario's four route pairs (utils.py:178), ego = combo mod 2.  Ego state,
previous input and the opponent's constant-acceleration forecast
(constant_acceleration_model.py:69-71) are drawn as section 8d specifies; the forecast is
mapped to x,y with routes.frenet2global and passed through routes.filter_preds."""
import numpy as np

from . import routes as R


def make_batch(B, N=20, dt=0.1, seed=2026, dtype=np.float32, offset=0):
    """-> dict of host arrays: x0[B,7], u_prev[B,2], kparams[B,3], flags[B] (uint32),
    obs_xy[B,1,2,N+1], tv_sv[B,2], enc[B,2], plus bookkeeping (sc, ego/opp route ids).
    `offset` shifts the batch index (rank r of an N-way shard passes offset = r*B)."""
    rng = np.random.default_rng(seed) if offset == 0 else np.random.default_rng([seed, offset])
    b = np.arange(offset, offset + B)
    combo = b % 64
    sc = combo // 8 + 1
    rot = (combo // 2) % 4
    ego_i = combo % 2
    ego_rid = np.empty(B, dtype=np.int64)
    opp_rid = np.empty(B, dtype=np.int64)
    enc = np.empty((B, 2))
    for i in range(B):
        pair = R.SCENARIO_ROUTES[sc[i] - 1][rot[i]]
        e = R.scenario_encoding_sign(pair, int(sc[i]))
        ego_rid[i] = R.ROUTE_ID[pair[ego_i[i]]]
        opp_rid[i] = R.ROUTE_ID[pair[1 - ego_i[i]]]
        enc[i] = (e[ego_i[i]], e[1 - ego_i[i]])

    s0 = rng.uniform(0, 45, B)
    v0 = rng.uniform(0.5, 4.5, B)
    ey0 = rng.uniform(-0.1, 0.1, B)
    ep0 = rng.uniform(-0.05, 0.05, B)
    a_prev = rng.uniform(-1, 1, B)
    df_prev = rng.uniform(-0.1, 0.1, B)
    s_op = rng.uniform(0, 45, B)
    v_op = rng.uniform(0, 5, B)
    a_op = rng.uniform(-1, 1, B)

    th = R.psi_ref(ego_rid, s0)
    xy = R.frenet2global(ego_rid, s0)
    x0 = np.empty((B, 7))
    x0[:, 0] = xy[:, 0] - ey0 * np.sin(th)
    x0[:, 1] = xy[:, 1] + ey0 * np.cos(th)
    x0[:, 2], x0[:, 3], x0[:, 4], x0[:, 5] = s0, ey0, ep0, v0
    x0[:, 6] = th + ep0
    flags = np.where(R.TABLES['abs_heading'][ego_rid], 1, 0).astype(np.uint32)
    kp = R.kparams(ego_rid)

    # opponent: constant acceleration in s (constant_acceleration_model.py:69-71)
    obs = np.empty((B, 1, 2, N + 1))
    s, v = s_op.copy(), v_op.copy()
    obs[:, 0, :, 0] = R.frenet2global(opp_rid, s)
    for k in range(N):
        s = s + (v * dt + 0.5 * a_op * dt ** 2)
        v = np.clip(v + a_op * dt, -2.0, 20.0)         # fourwayint.yaml:23-24
        obs[:, 0, :, k + 1] = R.frenet2global(opp_rid, s)
    obs = R.filter_preds(x0[:, 0:2], x0[:, 6], obs)
    tv_sv = np.stack([s, v], axis=-1)                   # last raw prediction (mpc.py:330)

    out = dict(x0=x0, u_prev=np.stack([a_prev, df_prev], -1), kparams=kp, obs_xy=obs, tv_sv=tv_sv, enc=enc)
    out = {k: np.ascontiguousarray(v.astype(dtype)) for k, v in out.items()}
    out['flags'] = flags
    out['sc'] = sc
    out['ego_rid'] = ego_rid
    out['opp_rid'] = opp_rid
    return out
