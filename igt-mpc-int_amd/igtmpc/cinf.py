"""Maximal control-invariant set of the (v, a) double integrator -- the terminal set of
the reference planner (mpc.py:88-104 builds it with the `polytope` library through
utils.Cinf / precursor / minkowski_sum, utils.py:588-627; mpc.py:177-180 applies it to
(v_{N-1}, a_{N-1})).

`polytope` is not part of this stack, and the sets here are 2-D, so this is an own
convex-polygon implementation of the same fixed point

    Omega_0 = X,    Omega_{k+1} = Pre(Omega_k) & Omega_k,
    Pre(Omega) = { x : A x in Omega (+) (-B U) }

with X = {v<=5, a<=3, v>=-1, a>=-4}, U = {|da| <= dt*jerk}, A = [[1,dt],[0,1]], B = [0,1]^T.
The result is returned as unit-normal half-planes  A_h [F,2] (v,a) <= b_h [F].
PARITY UNPINNED against polytope (absent offline); tests pin it with invariance /
maximality properties instead (tests/test_properties.py / tests/test_host_logic.py)."""
import numpy as np


def _hull(points):
    """Convex hull, counter-clockwise, Andrew monotone chain; drops collinear points.
    Coordinates are snapped to a 1e-12 grid first: vertices produced by clipping sit at
    -1 +- 1ulp, and an exact-arithmetic chain would otherwise order (and then discard)
    points of a vertical edge by that noise."""
    pts = sorted(set((round(float(x), 12) + 0.0, round(float(y), 12) + 0.0) for x, y in points))
    if len(pts) <= 2:
        return np.array(pts)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 1e-15:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 1e-15:
            upper.pop()
        upper.append(p)
    return np.array(lower[:-1] + upper[:-1])


def _clip(poly, n, b):
    """Sutherland-Hodgman: keep the part of convex polygon `poly` with n.x <= b."""
    out = []
    m = len(poly)
    for i in range(m):
        p, q = poly[i], poly[(i + 1) % m]
        dp, dq = n @ p - b, n @ q - b
        if dp <= 0:
            out.append(p)
        if (dp < 0 < dq) or (dq < 0 < dp):
            t = dp / (dp - dq)
            out.append(p + t * (q - p))
    return np.array(out)


def _halfplanes(poly):
    """CCW vertices -> unit outward normals and offsets."""
    A, b = [], []
    m = len(poly)
    for i in range(m):
        p, q = poly[i], poly[(i + 1) % m]
        e = q - p
        L = np.hypot(*e)
        if L < 1e-12:
            continue
        n = np.array([e[1], -e[0]]) / L
        A.append(n)
        b.append(n @ p)
    return np.array(A), np.array(b)


def _merge_close(poly, tol):
    keep = []
    for p in poly:
        if not keep or np.hypot(*(p - keep[-1])) > tol:
            keep.append(p)
    if len(keep) > 1 and np.hypot(*(keep[0] - keep[-1])) <= tol:
        keep.pop()
    return np.array(keep)


def _canonical(poly):
    """Rotate the CCW vertex list so it starts at the lexicographically smallest vertex."""
    k = min(range(len(poly)), key=lambda i: (poly[i][0], poly[i][1]))
    return np.roll(poly, -k, axis=0)


def _prune_collinear(poly, tol):
    """Drop vertices closer than tol to the chord of their neighbours (a vertex that sits on
    an edge flips in and out of the hull with rounding and would keep the fixed point from
    being recognised)."""
    poly = list(poly)
    changed = True
    while changed and len(poly) > 3:
        changed = False
        for i in range(len(poly)):
            p, q, r = poly[i - 1], poly[i], poly[(i + 1) % len(poly)]
            e = r - p
            L = np.hypot(*e)
            if L > 0 and abs(e[0] * (q[1] - p[1]) - e[1] * (q[0] - p[0])) / L < tol:
                poly.pop(i)
                changed = True
                break
    return np.array(poly)


def _same_set(P, Q, tol):
    """Hausdorff-style equality of two convex polygons: each one's vertices satisfy the
    other's half-planes within tol."""
    for U, V in ((P, Q), (Q, P)):
        A, b = _halfplanes(V)
        if (U @ A.T - b).max() > tol:
            return False
    return True


def control_invariant_set(dt=0.1, jerk=0.9, v_lo=-1.0, v_hi=5.0, a_lo=-4.0, a_hi=3.0, tol=1e-9, max_iter=500):
    """-> (A_h[F,2], b_h[F], vertices[F,2], iterations)."""
    Am = np.array([[1.0, dt], [0.0, 1.0]])
    Ainv = np.linalg.inv(Am)
    r = dt * jerk                                         # mpc.py:97-100
    X = np.array([[v_lo, a_lo], [v_hi, a_lo], [v_hi, a_hi], [v_lo, a_hi]])      # mpc.py:88-95 (CCW)
    XA, Xb = _halfplanes(X)
    omega = _canonical(X)
    for it in range(1, max_iter + 1):
        # Omega (+) (-B U): sweep along the a-axis by +-r (utils.py:603-627)
        swept = _hull(np.concatenate([omega + [0.0, r], omega - [0.0, r]]))
        # pre-image under A (utils.py:601: Polytope(tmp.A @ A, tmp.b))
        pre = swept @ Ainv.T
        # intersect with Omega (utils.py:591)
        nxt = pre
        OA, Ob = _halfplanes(omega)
        for n, b in zip(OA, Ob):
            nxt = _clip(nxt, n, b)
            if len(nxt) < 3:
                raise RuntimeError('control-invariant set collapsed')
        nxt = _canonical(_prune_collinear(_merge_close(_hull(nxt), 1e-10), 1e-10))
        if _same_set(nxt, omega, tol):
            omega = nxt
            break
        omega = nxt
    else:
        raise RuntimeError('control-invariant set did not converge')
    A_h, b_h = _halfplanes(omega)
    return A_h, b_h, omega, it


_CACHE = {}


def cinf_halfplanes(dt=0.1, jerk=0.9, **kw):
    key = (dt, jerk, tuple(sorted(kw.items())))
    if key not in _CACHE:
        A, b, _, _ = control_invariant_set(dt=dt, jerk=jerk, **kw)
        _CACHE[key] = (A, b)
    return _CACHE[key]
