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
