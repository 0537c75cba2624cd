"""Boundary data types of the path: VehicleState / VehicleAction / VehicleReference.

Same constructor (one dict) and attribute names as the reference's attribute bags
(common/VehicleState.py:3-23, VehicleAction.py:3-10, VehicleReference.py:3-27), so objects
built by an evaluate.py-style driver pass straight through.  Implemented once over a field
table instead of three hand-written classes."""


class _Bag:
    _required = ()
    _optional_group = ()      # present together or not at all

    def __init__(self, values: dict):
        self.update(values)

    def update(self, values: dict):
        for k in self._required:
            setattr(self, k, values[k])
        if self._optional_group and self._optional_group[0] in values:
            for k in self._optional_group:
                setattr(self, k, values[k])

    def as_dict(self):
        return {k: getattr(self, k) for k in self._required + self._optional_group if hasattr(self, k)}

    def __repr__(self):
        return f'{type(self).__name__}({self.as_dict()})'


class VehicleState(_Bag):
    """x, y, heading, v and -- when given -- the Frenet triple s, ey, epsi (VehicleState.py:10-13)."""
    _required = ('x', 'y', 'heading', 'v')
    _optional_group = ('s', 'ey', 'epsi')


class VehicleAction(_Bag):
    """a (acceleration), df (front steering angle)."""
    _required = ('a', 'df')


class VehicleReference(_Bag):
    """Full Frenet + Cartesian state with the curvature function K of the vehicle's route
    (K is a callable for ego states, evaluate.py:418; None for predictions,
    constant_acceleration_model.py:35)."""
    _required = ('x', 'y', 'heading', 'v', 's', 'K', 'ey', 'epsi')

    def get_state_array(self):
        return [self.x, self.y, self.s, self.ey, self.epsi, self.v]      # no heading (VehicleReference.py:26-27)

    def state7(self):
        """planner order [x, y, s, ey, epsi, v, psi] (mpc.py:163)."""
        return [self.x, self.y, self.s, self.ey, self.epsi, self.v, self.heading]


class Curvature:
    """K(s) of a route as the hot path consumes it: Kv on [b0, b1), 0 elsewhere
    (= ca.pw_const(s,[b0,b1],[0,Kv,0]), mpc.py:183-200).  Callable like the reference's
    casadi Function, and carries the three numbers the kernels need."""

    def __init__(self, b0=float('inf'), b1=float('inf'), Kv=0.0):
        self.kparams = (float(b0), float(b1), float(Kv))

    def __call__(self, s):
        b0, b1, kv = self.kparams
        return (kv if s >= b0 else 0.0) - (kv if s >= b1 else 0.0)

    @classmethod
    def from_route(cls, route):
        from . import routes as R
        return cls(*R.kparams(R.ROUTE_ID[route]))

    @classmethod
    def from_reference(cls, ref_K, route, road_dim=(11.4, 50.0), ds_right=8.6):
        """The construction of mpc.py:183-200 from a reference-path curvature array."""
        import numpy as np
        K = np.asarray(ref_K, dtype=float)
        nz = np.nonzero(K)[0]
        if len(nz) == 0:
            return cls()
        W, L = road_dim
        radius = float(np.max(np.abs(1.0 / K[nz])))
        b0 = (L - W) / 2 if route in ('12', '23', '34', '41') else (L - W) / 2 - ds_right
        return cls(b0, b0 + radius * np.pi / 2, float(K[nz[0]]))

    @classmethod
    def from_callable(cls, K, s_max=200.0, step=0.05):
        """Recovers (b0, b1, Kv) from ANY callable curvature K(s) of the reference's shape -- piecewise constant, 0
        outside one interval [b0, b1) (mpc.py:183-200; evaluate.py:384-402 builds it as a casadi Function) -- by
        sampling it on a grid and bisecting each jump to the last representable s (the break-points come out exactly
        as the float64 numbers the callable compares against).  The result is cached on the callable."""
        cached = getattr(K, '_igt_curvature', None)
        if cached is not None:
            return cached
        f = lambda s: float(K(s))
        n = int(round(s_max / step))
        vals = [f(i * step) for i in range(n + 1)]
        jumps = [i for i in range(n) if vals[i] != vals[i + 1]]
        if not jumps:
            if vals[0] != 0.0:
                raise ValueError('curvature callable is a non-zero constant: not a route of the four-way intersection')
            out = cls()
        else:
            if len(jumps) > 2 or vals[0] != 0.0 or (len(jumps) == 2 and vals[-1] != 0.0):
                raise ValueError('curvature callable is not 0 / Kv / 0 piecewise constant')
            bps = []
            for i in jumps:
                lo, hi, v_lo = i * step, (i + 1) * step, vals[i]
                while True:                       # invariant: K(lo) == v_lo != K(hi); ends at adjacent doubles
                    mid = 0.5 * (lo + hi)
                    if mid <= lo or mid >= hi:
                        break
                    if f(mid) == v_lo:
                        lo = mid
                    else:
                        hi = mid
                bps.append(hi)                    # first s with the new value: K(s) = Kv for s >= b0
            kv = vals[jumps[0] + 1]
            out = cls(bps[0], bps[1] if len(bps) == 2 else float('inf'), kv)
        try:
            K._igt_curvature = out
        except (AttributeError, TypeError):
            pass
        return out
