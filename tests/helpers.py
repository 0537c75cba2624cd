"""Shared parity harness (tests only).  The oracle is the float64 numpy restatement in
oracle/np_oracle.py, fed the SAME inputs as the device (float32 inputs are up-cast
exactly), so differences are the kernel's arithmetic only."""
import numpy as np

import np_oracle as O

REL_TOL = 1e-5       # BASELINE.json north_star: 1e-5 relative, |ref| floored at 1
F32_EPS = 1e-7       # f32 comparisons: width of the threshold / break-point set-aside (tools/f32_margin_probe.py)
F32_TIE = 1e-6       # ... and of the cost near-tie set-aside


def rel_err(got, ref):
    """|got-ref| / max(1,|ref|), elementwise."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(got - ref) / np.maximum(1.0, np.abs(ref))


def oracle_params(solver):
    p = solver.params
    return O.Params(N=p.N, dt=p.dt, n_rk4=p.n_rk4, l_r=p.l_r, l_f=p.l_f, v_min=p.v_min, v_max=p.v_max,
                    a_min=p.a_min, a_max=p.a_max, df_max=p.df_max, jerk=p.jerk_limit,
                    steer_rate=p.steer_rate_limit, ey_lim=p.ey_lim, d_min=p.d_min, w_u=p.w_u,
                    feas_tol=p.feas_tol)


def oracle_solve(batch, P, C=256, cinf=None, U=None):
    f = lambda k: np.asarray(batch[k], dtype=np.float64)
    A, b = (None, None) if cinf is None else cinf
    return O.solve_batch(f('x0'), f('u_prev'), f('kparams'), batch['flags'], f('obs_xy'), A, b, P, C=C, U=U,
                         return_all=True)


def ambiguous_mask(ref, P, eps_margin=2e-5, eps_cost=2e-5, eps_bp=2e-5, bp=None, ties=True):
    """Scenarios whose arg-min is decided inside float32 noise (documented in DESIGN.md):
    some candidate that could win sits within eps of a constraint threshold, or the best
    two feasible costs are closer than eps, or a stage argument of a contender lies
    within eps of a curvature break-point."""
    J, g, feas = ref['J'], ref['g'], ref['feas']
    Jm = np.where(feas, J, np.inf)
    best = Jm.min(axis=1)
    contender = J <= (best[:, None] + eps_cost)             # could win if verdict flipped
    near_thr = np.abs(g - P.feas_tol) < eps_margin
    amb = (contender & near_thr).any(axis=1)
    if ties:
        # the runner-up that matters is the best candidate whose CONTROLS differ from the winner's: candidates with the
        # same control sequence (targets clipped to the same envelope / box) are the same arithmetic on the same numbers,
        # tie exactly on both sides, and the lowest index wins on both sides (test_exact_ties_across_slices_resolve_to_the_lowest_index)
        rival = Jm
        if 'U' in ref:
            Uw = ref['U'][np.arange(J.shape[0]), np.argmin(Jm, axis=1)]
            rival = np.where((ref['U'] == Uw[:, None]).all(axis=(-1, -2)), np.inf, Jm)
        else:
            rival = np.sort(Jm, axis=1)[:, 1:2]
        with np.errstate(invalid='ignore'):
            amb |= (rival.min(axis=1) - best) < eps_cost
    if bp is not None:
        amb |= (contender & (bp < eps_bp)).any(axis=1)
    return amb


def verdict_margins(X, U, obs_xy, cinf_A, cinf_b, P):
    """Worst signed violation PER VERDICT FAMILY (np_oracle.constraint_violation's bits 0, 1, 3, 4, 5; -inf where a family
    does not apply): [..., 7].  Lets a test say that a differing verdict bit sits on its threshold."""
    N = U.shape[-1]
    shape = np.broadcast_shapes(X.shape[:-2], U.shape[:-2])
    g = np.full(shape + (7,), -np.inf)
    v = X[..., O.IV, :N]
    g[..., 0] = np.max(np.maximum(P.v_min - v, v - P.v_max), axis=-1)
    a, d = U[..., 0, :], U[..., 1, :]
    g[..., 1] = np.max(np.maximum(np.maximum(P.a_min - a, a - P.a_max), np.maximum(-P.df_max - d, d - P.df_max)), axis=-1)
    g[..., 3] = np.max(np.abs(X[..., O.IEY, :]) - P.ey_lim, axis=-1)
    if cinf_A is not None and len(cinf_b):
        t = cinf_A[:, 0] * X[..., O.IV, N - 1, None] + cinf_A[:, 1] * U[..., 0, N - 1, None] - cinf_b
        g[..., 4] = np.max(t, axis=-1)
    if obs_xy is not None and obs_xy.shape[-3] > 0:
        dx = X[..., None, O.IX, 1:] - obs_xy[..., :, 0, 1:]
        dy = X[..., None, O.IY, 1:] - obs_xy[..., :, 1, 1:]
        g[..., 5] = np.max(P.d_min ** 2 - (dx * dx + dy * dy), axis=(-1, -2))
    return g
