"""Four-way-intersection route geometry as arrays (host side, numpy).

The reference keeps this knowledge in three places: ReferenceGen.py:41-201 (reference
paths -> curvature value, radii), mpc.py:183-200 (curvature break-points) and
utils.py:532-586 `frenet2global` (s -> x,y for the 12 routes).  Here it is one table per
route, indexed by route id, so whole batches are mapped at once.

Numbers that come out of ReferenceGenerator (Kv, radius, headings, the frozen end
coordinates xN/yN used after the arc) are data: data/route_constants.json was produced by
running the reference (tests/golden/make_golden.py)."""
import json
import os

import numpy as np

ROUTES = ['12', '13', '14', '21', '23', '24', '31', '32', '34', '41', '42', '43']   # utils.py:394-399 order
ROUTE_ID = {r: i for i, r in enumerate(ROUTES)}
LEFT = ['12', '23', '34', '41']
RIGHT = ['14', '21', '32', '43']
STRAIGHT = ['13', '24', '31', '42']
ABS_HEADING = ['32', '41']        # mpc.py:231, 250, 273, 282

ROAD_LENGTH = 50.0                # fourwayint.yaml:5
ROAD_WIDTH = 11.4                 # fourwayint.yaml:3
CA_RADIUS = 2.8                   # fourwayint.yaml:9

# utils.py:178 (get_route_from_scenario) -- scenario 1..8 -> four route pairs (sets in the
# reference; sorted here so the agent order does not depend on PYTHONHASHSEED)
SCENARIO_ROUTES = [
    [('13', '23'), ('24', '34'), ('31', '41'), ('12', '42')],
    [('12', '41'), ('12', '23'), ('23', '34'), ('34', '41')],
    [('13', '24'), ('24', '31'), ('31', '42'), ('13', '42')],
    [('12', '32'), ('23', '43'), ('14', '34'), ('21', '41')],
    [('13', '43'), ('14', '24'), ('21', '31'), ('32', '42')],
    [('13', '41'), ('12', '24'), ('23', '31'), ('34', '42')],
    [('12', '34'), ('23', '41'), ('12', '34'), ('23', '41')],
    [('12', '31'), ('23', '42'), ('13', '34'), ('24', '41')],
]

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'route_constants.json')) as _f:
    CONSTANTS = json.load(_f)

_L, _W = ROAD_LENGTH, ROAD_WIDTH
# origin -> (start point of s, unit tangent)
_ORIGIN = {'1': ((0.0, CA_RADIUS), (1.0, 0.0)),
           '2': ((_L / 2 - _W / 2 + CA_RADIUS, (_L + _W) / 2), (0.0, -1.0)),
           '3': ((_L, _W - CA_RADIUS), (-1.0, 0.0)),
           '4': ((_L / 2 + _W / 2 - CA_RADIUS, (_W - _L) / 2), (0.0, 1.0))}


def _tables():
    n = len(ROUTES)
    T = dict(p0=np.zeros((n, 2)), t=np.zeros((n, 2)), c=np.zeros((n, 2)), b0=np.full(n, np.inf),
             b1=np.full(n, np.inf), R=np.ones(n), Kv=np.zeros(n), end=np.zeros((n, 2)),
             h0=np.zeros(n), hN=np.zeros(n), turn=np.zeros(n), abs_heading=np.zeros(n, dtype=bool))
    for i, r in enumerate(ROUTES):
        k = CONSTANTS[r]
        p0, t = _ORIGIN[r[0]]
        T['p0'][i] = p0
        T['t'][i] = t
        T['h0'][i] = k['heading0']
        T['hN'][i] = k['headingN']
        T['end'][i] = (k['xN'], k['yN'])
        T['abs_heading'][i] = r in ABS_HEADING
        if not k['straight']:
            turn = 1.0 if r in LEFT else -1.0
            T['turn'][i] = turn
            T['c'][i] = (-t[1] * turn, t[0] * turn)        # side the arc bends to
            T['b0'][i], T['b1'][i], T['R'][i], T['Kv'][i] = k['b0'], k['b1'], k['radius'], k['Kv']
    return T


TABLES = _tables()


def kparams(route_ids):
    """(b0, b1, Kv) per route id -- the curvature function of mpc.py:183-200;
    straight routes get (inf, inf, 0)."""
    rid = np.asarray(route_ids)
    return np.stack([TABLES['b0'][rid], TABLES['b1'][rid], TABLES['Kv'][rid]], axis=-1)


def frenet2global(route_ids, s):
    """Centre-line point at arc length s (utils.py:532-586), vectorised over (route id, s).
    Before the arc: start + s*t.  On the arc (b0 <= s <= b1): circular.  After it the
    reference freezes the along-coordinate at the end value of its reference path
    (ref['x'][-1] / ref['y'][-1]) and advances the cross-coordinate by s - b1 + R."""
    rid = np.asarray(route_ids)
    s = np.asarray(s, dtype=np.float64)
    T = TABLES
    p0, t, c = T['p0'][rid], T['t'][rid], T['c'][rid]
    b0, b1, R = T['b0'][rid], T['b1'][rid], T['R'][rid]
    straight = T['turn'][rid] == 0
    pre = (s < b0) | straight
    post = (s > b1) & ~straight
    with np.errstate(invalid='ignore'):
        phi = np.where(pre, 0.0, (s - np.where(straight, 0.0, b0)) / R)
    along_arc = np.where(straight, s, b0 + R * np.sin(phi))
    cross_arc = R * (1 - np.cos(phi))
    along = np.where(pre, s, along_arc)
    cross = np.where(pre, 0.0, np.where(post, s - b1 + R, cross_arc))
    xy = p0 + along[..., None] * t + cross[..., None] * c
    # frozen along-coordinate after the arc
    end_along = (T['end'][rid] * np.abs(t)).sum(-1)
    xy_post = end_along[..., None] * np.abs(t) + (p0 * np.abs(c)) + cross[..., None] * c
    return np.where(post[..., None], xy_post, xy)


def psi_ref(route_ids, s):
    """Reference heading at s: ca.pw_lin(s, [0,b0,b1,1000], [h0,h0,hN,hN])
    (constant_acceleration_model.py:46-66); constants already carry |.| for '32','41'."""
    rid = np.asarray(route_ids)
    s = np.asarray(s, dtype=np.float64)
    T = TABLES
    b0, b1, h0, hN = T['b0'][rid], T['b1'][rid], T['h0'][rid], T['hN'][rid]
    straight = T['turn'][rid] == 0
    with np.errstate(invalid='ignore'):
        w = np.clip((s - b0) / np.where(straight, 1.0, b1 - b0), 0.0, 1.0)
    return np.where(straight, h0, h0 + w * (hN - h0))


def filter_preds(ego_xy0, ego_heading, obs_xy):
    """utils.py:365-388: an obstacle whose current position is behind the ego
    ((p_obs - p_ego) . (cos psi, sin psi) < 0) is moved to (-20,-20) for the whole
    horizon.  ego_xy0[B,2], ego_heading[B], obs_xy[B,n_obs,2,N+1] -> filtered copy."""
    obs = np.array(obs_xy, dtype=np.float64, copy=True)
    d = obs[:, :, :, 0] - np.asarray(ego_xy0)[:, None, :]
    dot = d[..., 0] * np.cos(ego_heading)[:, None] + d[..., 1] * np.sin(ego_heading)[:, None]
    behind = dot < 0
    obs[behind] = -20.0
    return obs


def scenario_encoding_sign(routes, sc):
    """utils.py:84-139 scenario_index: returns (e_0, e_1) = (+sc,-sc) or (-sc,+sc)."""
    m = sc
    vh1, vh2 = routes[0][0], routes[1][0]
    if m == 1:
        if routes[0] == '42':
            vh1 = '0'
        elif routes[1] == '42':
            vh2 = '0'
    elif m == 2:
        if routes[0] == '12' and routes[1] == '41':
            vh1 = '5'
        elif routes[1] == '12' and routes[0] == '41':
            vh2 = '5'
    elif m == 3:
        if routes[0] == '13' and routes[1] == '42':
            vh2 = '0'
        elif routes[1] == '13' and routes[0] == '42':
            vh1 = '0'
    if m < 4 or m == 7:
        first = int(vh1) < int(vh2)
    elif m in (4, 6, 8):
        first = routes[0] in LEFT
    elif m == 5:
        first = routes[0] in STRAIGHT
    else:
        raise ValueError('Scenario not found')
    return (m, -m) if first else (-m, m)


def scenario_of(routes):
    """utils.py:141-169: which of the 8 scenarios a route pair belongs to."""
    key = tuple(sorted(routes))
    for sc, pairs in enumerate(SCENARIO_ROUTES, start=1):
        if key in [tuple(sorted(p)) for p in pairs]:
            return sc
    raise ValueError('Scenario not found')
