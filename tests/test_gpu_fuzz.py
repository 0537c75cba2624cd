"""-m gpu : randomised configurations of the float64 entry points against the numpy oracle.

The other parity tests run the reference's configuration (N = 20 or 40, 256 candidates, 4 RK4 sub-steps, dt = 0.1) at
many batch sizes; the kernels, however, are templates over much more than that -- the build for any n_rk4 next to the one
for 4, high-order offset polynomials for coarse discretisations, 1 to 64 units per scenario, steering tables that fit LDS
or do not, one or more obstacles, trajectories kept by the search pass or re-rolled, one wave per unit or queues.  Here a
seeded generator draws the configuration -- horizon, discretisation, dt, candidate count, family, refinement passes,
obstacle count, limits, batch size -- and every draw must give the oracle's answer: status and arg-min exactly wherever
the decision is not inside 1e-9 of a threshold, tie or break-point, trajectories and costs to 1e-9."""
import numpy as np
import pytest

import np_oracle as O
from helpers import F32_EPS, F32_TIE, REL_TOL, ambiguous_mask, oracle_params, rel_err


def _draw(seed):
    if seed >= 200:
        return _draw_ext(seed)
    rng = np.random.default_rng([2026, seed])
    cfg = dict(
        N=int(rng.choice([4, 7, 12, 20, 33, 40])),
        n_rk4=int(rng.choice([1, 2, 3, 4, 4, 6])),
        dt=float(rng.choice([0.05, 0.1, 0.1, 0.2])),
        C=int(rng.choice([64, 256, 256, 1024])),
        cand=str(rng.choice(['lattice', 'ramp_hold', 'track', 'track'])),
        n_obs=int(rng.choice([0, 1, 1, 2])),
        B=int(rng.choice([1, 3, 8, 17, 40, 70, 300])),
        terminal=bool(rng.random() < 0.6),
    )
    cfg['refine'] = int(rng.choice([0, 0, 1, 2])) if cfg['cand'] != 'lattice' else 0
    cfg['warm'] = bool(cfg['cand'] != 'lattice' and rng.random() < 0.4)       # igt_solve_batch_ws_*: two thirds of the batch
    cfg['net'] = int(rng.choice([0, 0, 0, 1, 3])) if cfg['n_obs'] == 1 else 0    # terminal value network V_GT_sc1 / sc3 (0: off)
    if cfg['B'] * cfg['C'] * cfg['N'] * cfg['n_rk4'] > 2.5e7:                 # keeps the numpy oracle to about a second
        cfg['B'] = 17
    lim = {}
    if rng.random() < 0.5:
        lim['v_max'] = float(rng.choice([4.0, 6.0]))
    if rng.random() < 0.5:
        lim['a_max'] = float(rng.choice([2.0, 3.5]))
    if rng.random() < 0.3:
        lim['ey_lim'] = 0.35
    if rng.random() < 0.3:
        lim['d_min'] = float(rng.choice([4.0, 7.0]))
    if rng.random() < 0.3:
        lim['w_u'] = 0.3
    if cfg['cand'] == 'track' and rng.random() < 0.5:
        lim['track_env'] = float(rng.choice([0.0, 0.5]))
    if cfg['cand'] == 'track' and rng.random() < 0.3:
        lim['track_vcap'] = 0.0
    cfg['limits'] = lim
    return cfg


def _draw_ext(seed):
    """The ranges include/igtmpc.h advertises beyond the reference's configuration: horizons up to IGT_MAX_N = 64 (with C = 256 the
    slice's steering table fits LDS, with C = 64 it does not: both paths), up to IGT_MAX_OBS = 4 obstacles, batches either side
    of the live-row units' bound."""
    rng = np.random.default_rng([4026, seed])
    cfg = dict(
        N=int(rng.choice([48, 64, 64])),
        n_rk4=int(rng.choice([2, 4, 4])),
        dt=float(rng.choice([0.05, 0.1])),
        C=int(rng.choice([64, 256, 256])),
        cand=str(rng.choice(['lattice', 'ramp_hold', 'track'])),
        n_obs=int(rng.choice([2, 3, 4, 4])),
        B=int(rng.choice([5, 40, 130, 1100])),
        terminal=bool(rng.random() < 0.6),
    )
    cfg['refine'] = int(rng.choice([0, 0, 1])) if cfg['cand'] != 'lattice' else 0
    cfg['warm'] = bool(cfg['cand'] != 'lattice' and rng.random() < 0.4)
    cfg['net'] = 0
    if cfg['B'] * cfg['C'] * cfg['N'] * cfg['n_rk4'] * (1 + cfg['refine']) > 9e7:
        cfg['B'] = 130 if cfg['C'] == 256 else 1100
    lim = {}
    if rng.random() < 0.5:
        lim['v_max'] = float(rng.choice([4.0, 6.0]))
    if rng.random() < 0.3:
        lim['d_min'] = float(rng.choice([4.0, 7.0]))
    if cfg['cand'] == 'track' and rng.random() < 0.5:
        lim['track_env'] = float(rng.choice([0.0, 0.5]))
    if cfg['cand'] == 'track' and rng.random() < 0.3:
        lim['track_vcap'] = 0.0
    cfg['limits'] = lim
    return cfg


N_SEEDS = 40
F32_SEEDS = range(100, 118)
EXT_SEEDS = range(200, 214)
EXT_F32_SEEDS = range(214, 220)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(N_SEEDS))
def test_random_configuration_matches_oracle(seed, golden_dir):
    _run(seed, golden_dir, 'f64')


@pytest.mark.gpu
@pytest.mark.parametrize('seed', F32_SEEDS)
def test_random_configuration_matches_oracle_f32(seed, golden_dir):
    """The float32 entry points on the same kind of draws, at BASELINE.json's 1e-5 with the float32 set-asides of
    test_gpu_parity.py (threshold / break-point within 1e-7, near-ties; value network: 2e-5)."""
    _run(seed, golden_dir, 'f32')


@pytest.mark.gpu
@pytest.mark.parametrize('seed', EXT_SEEDS)
def test_random_configuration_at_the_advertised_limits_matches_oracle(seed, golden_dir):
    _run(seed, golden_dir, 'f64')


@pytest.mark.gpu
@pytest.mark.parametrize('seed', EXT_F32_SEEDS)
def test_random_configuration_at_the_advertised_limits_matches_oracle_f32(seed, golden_dir):
    _run(seed, golden_dir, 'f32')


def _run(seed, golden_dir, dtype):
    import igtmpc
    from igtmpc.cinf import cinf_halfplanes
    from igtmpc.scenarios import make_batch
    cfg = _draw(seed)
    N, dt, C, B = cfg['N'], cfg['dt'], cfg['C'], cfg['B']
    f32 = dtype == 'f32'
    npdt = np.float32 if f32 else np.float64
    tol, utol = (REL_TOL, 1e-7 if cfg['cand'] != 'track' else REL_TOL) if f32 else (1e-9, 1e-12)
    eps, tie = ((2e-5, 2e-5) if cfg['net'] else (F32_EPS, F32_TIE)) if f32 else (1e-9, 1e-9)
    b = make_batch(max(B, 8), N=N, dt=dt, seed=100 + seed, dtype=npdt)
    b = {k: np.ascontiguousarray(v[:B]) for k, v in b.items() if isinstance(v, np.ndarray) and len(v) >= B}
    obs = b['obs_xy']                                                  # [B, 1, 2, N+1]
    if cfg['n_obs'] == 0:
        obs = np.zeros((B, 0, 2, N + 1), dtype=npdt)
    elif cfg['n_obs'] >= 2:                                            # further vehicles, 9 m apart behind the first along its path
        more = [obs]
        for m in range(1, cfg['n_obs']):
            lag = obs.copy()
            lag[:, 0, 0, :] -= 9.0 * m * np.cos(0.3 * np.arange(B))[:, None]
            lag[:, 0, 1, :] -= 9.0 * m * np.sin(0.3 * np.arange(B))[:, None]
            more.append(lag)
        obs = np.ascontiguousarray(np.concatenate(more, axis=1))
    rng = np.random.default_rng([7, seed])
    flags, u_prev, u_ws = b['flags'], b['u_prev'], None
    if cfg['warm']:       # previous solution = some lattice candidate of the scenario, shifted by one step (utils.py:354-363)
        P0 = O.Params(N=N, dt=dt)
        prev = O.candidates_lattice(b['u_prev'], P0)[np.arange(B), (np.arange(B) * 37 + seed) % 256]
        u_ws = np.ascontiguousarray(O.shift_controls(prev).astype(npdt))
        u_prev = np.ascontiguousarray(prev[:, :, 0].astype(npdt))
        flags = flags | np.where(np.arange(B) % 3 != 0, 2, 0).astype(np.uint32)
    net, extra = None, ()
    if cfg['net']:
        v = np.load(f'{golden_dir}/value_net_golden.npz')
        layers, i = [], 0
        while f"sc{cfg['net']}_W{i}" in v:
            layers.append((v[f"sc{cfg['net']}_W{i}"], v[f"sc{cfg['net']}_b{i}"]))
            i += 1
        net = dict(layers=layers, Wn=np.eye(6) + 0.05 * rng.normal(size=(6, 6)),
                   mu_f=np.array([20.0, 2.5, 0.0, 0.0, 0.0, 0.0]) + 0.1 * rng.normal(size=6), sigma_t=float(rng.choice([1.0, 3.0, -2.0])),
                   mu_t=float(rng.normal()))
        extra = (b['tv_sv'], b['enc'])
    with igtmpc.BatchSolver(N=N, dt=dt, n_rk4=cfg['n_rk4'], C=C, n_obs=cfg['n_obs'], dtype=dtype, cand_mode=cfg['cand'],
                            refine_iters=cfg['refine'], cost_mode='value_net' if net else 'progress', **cfg['limits']) as s:
        P = oracle_params(s)
        cinf = cinf_halfplanes(dt=dt, jerk=s.params.jerk_limit) if cfg['terminal'] else (None, None)
        if cfg['terminal']:
            s.set_cinf(*cinf)
        if net:
            s.set_value_net(**net)
        tk = dict(ke=s.params.track_ke, span=s.params.track_span, blim=s.params.track_beta_lim, env=s.params.track_env,
                  vcap=s.params.track_vcap)
        got = s.solve(b['x0'], u_prev, b['kparams'], flags, obs, *extra, u_ws=u_ws)
        n_all = min(B, 4)
        allc = s.rollout_all(b['x0'][:n_all], u_prev[:n_all], b['kparams'][:n_all], flags[:n_all], obs[:n_all],
                             *[e[:n_all] for e in extra], u_ws=None if u_ws is None else u_ws[:n_all])
    f = lambda k: np.asarray(b[k], dtype=np.float64)
    o = obs.astype(np.float64) if cfg['n_obs'] else None
    u_prev = u_prev.astype(np.float64)
    u_ws = None if u_ws is None else u_ws.astype(np.float64)
    kw = dict(net=net, tv_sv=f('tv_sv'), enc=f('enc')) if net else {}
    if cfg['cand'] == 'lattice':
        passes = [O.solve_batch(f('x0'), u_prev, f('kparams'), flags, o, cinf[0], cinf[1], P, C=C, return_all=True, **kw)]
    else:
        passes = O.solve_batch_refined(f('x0'), u_prev, f('kparams'), flags, o, cinf[0], cinf[1], P, C=C,
                                       refine_iters=cfg['refine'], cand=cfg['cand'], track=tk, u_ws=u_ws, **kw)
    ref, first = passes[-1], passes[0]
    x0 = O.apply_flags(f('x0'), flags)[:, None, :]
    kp = f('kparams')[:, None, :]
    # every candidate of the first pass: controls, trajectories, verdicts (the refinement passes re-centre on a winner)
    if cfg['refine'] == 0:
        bp_all = O.breakpoint_distance(x0[:n_all], first['U'][:n_all], kp[:n_all], P)
        clear = bp_all > eps
        if f32:     # float32 error grows with the excursion: only roll-outs that stay near the lane are held to 1e-5 (every
                    # feasible one does: |e_y| <= ey_lim)
            clear &= (np.abs(first['X'][:n_all, :, 3, :]).max(axis=-1) <= 1.0) & (np.abs(first['X'][:n_all, :, 4, :]).max(axis=-1) <= 0.5)
        if not clear.any():
            clear[...] = False
        assert clear.sum() == 0 or rel_err(allc['U'][clear], first['U'][:n_all][clear]).max() <= utol, cfg
        # ... those that stay clear of the model's singularity 1 - K e_y = 0 (frenet.py:73): a wild candidate of a long horizon
        # drifts tens of metres off the lane, and next to the pole rounding differences are amplified without bound
        # (seed 31: min |1 - K e_y| = 2.7e-4, 0.25 relative).  Every FEASIBLE candidate has |e_y| <= ey_lim and is compared.
        pole = np.abs(1.0 - kp[:n_all, :, 2:3] * first['X'][:n_all, :, 3, :]).min(axis=-1) > 0.1
        fin = clear & pole & np.isfinite(first['X'][:n_all]).all(axis=(-1, -2))
        assert (first['feas'][:n_all] <= pole).all()
        assert fin.sum() == 0 or rel_err(allc['X'][fin], first['X'][:n_all][fin]).max() <= tol, cfg
        thr = fin & (np.abs(first['g'][:n_all] - P.feas_tol) > (1e-6 if f32 else 1e-9))
        assert ((allc['viol'] == 0) == first['feas'][:n_all])[thr].all(), cfg
    # the solve: a scenario is set aside when ANY pass decided it inside 1e-9 (a different winner re-centres the next pass)
    amb = np.zeros(B, dtype=bool)
    for r in passes:
        amb |= ambiguous_mask(r, P, eps, tie, eps, O.breakpoint_distance(x0, r['U'], kp, P))
    ok = ~amb
    assert (got['status'][ok] == ref['status'][ok]).all(), cfg
    assert (got['argmin'][ok] == ref['argmin'][ok]).all(), cfg
    sol = ok & (ref['status'] == 0)
    if sol.any():
        assert rel_err(got['x'][sol], ref['x'][sol]).max() <= tol, cfg
        assert rel_err(got['u'][sol], ref['u'][sol]).max() <= max(tol, utol), cfg
        assert rel_err(got['cost'][sol], ref['cost'][sol]).max() <= (2e-5 if f32 and cfg['net'] else tol), cfg
    bad = got['status'] == 1
    assert np.isnan(got['x'][bad]).all() and np.isinf(got['cost'][bad]).all() and (got['argmin'][bad] == -1).all(), cfg


def test_the_draws_cover_the_template_space():
    """The seeds above are only worth something if they reach the corners: both RK4 builds, coarse and fine steps,
    1 / 4 / 16 units per scenario, every family, refinement, warm starts, 0 / 1 / 2 obstacles, with and without the terminal
    set, both value-network architectures, batches either side of the kept-trajectory bound (units <= 1024 SIMDs)."""
    cfgs = [_draw(s) for s in range(N_SEEDS)]
    assert {c['net'] for c in cfgs} == {0, 1, 3} and any(c['warm'] for c in cfgs)
    assert any(c['net'] and c['cand'] == 'track' for c in cfgs) and any(c['net'] and c['refine'] for c in cfgs)
    assert any(c['B'] * c['C'] // 64 > 1024 for c in cfgs) and any(c['B'] * c['C'] // 64 <= 1024 and c['B'] > 8 for c in cfgs)
    f32 = [_draw(s) for s in F32_SEEDS]
    assert {c['cand'] for c in f32} == {'lattice', 'ramp_hold', 'track'} and any(c['net'] for c in f32) and any(c['warm'] for c in f32)
    assert {c['C'] for c in f32} == {64, 256, 1024} and any(c['n_rk4'] != 4 for c in f32) and any(c['refine'] for c in f32)
    assert {c['n_rk4'] == 4 for c in cfgs} == {True, False}
    assert {c['C'] for c in cfgs} == {64, 256, 1024}
    assert {c['cand'] for c in cfgs} == {'lattice', 'ramp_hold', 'track'}
    assert {c['n_obs'] for c in cfgs} == {0, 1, 2}
    assert {c['terminal'] for c in cfgs} == {True, False}
    assert any(c['refine'] > 0 for c in cfgs) and any(c['N'] > 20 for c in cfgs) and any(c['N'] < 12 for c in cfgs)
    assert any(c['n_rk4'] <= 2 and c['dt'] >= 0.1 for c in cfgs)          # the high-order offset polynomials
    assert any(c['B'] == 1 for c in cfgs) and any(c['B'] >= 17 for c in cfgs)
    ext = [_draw(s) for s in list(EXT_SEEDS) + list(EXT_F32_SEEDS)]
    assert {c['n_obs'] for c in ext} == {2, 3, 4} and {c['N'] for c in ext} == {48, 64} and {c['C'] for c in ext} == {64, 256}
    assert any(c['N'] == 64 and c['C'] == 64 for c in ext) and any(c['N'] == 64 and c['C'] == 256 for c in ext)     # table fits / does not
    assert any(c['B'] * c['C'] // 64 > 4096 for c in ext) and {c['cand'] for c in ext} == {'lattice', 'ramp_hold', 'track'}
    assert any(c['n_obs'] == 4 and c['N'] == 64 for c in ext)
