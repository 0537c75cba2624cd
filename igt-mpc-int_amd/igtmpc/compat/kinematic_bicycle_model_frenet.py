"""shim: `from kinematic_bicycle_model_frenet import KinematicBicycleModelFrenet` -> igtmpc.models (see compat/README.md)"""
from igtmpc.models import KinematicBicycleModelFrenet  # noqa: F401
