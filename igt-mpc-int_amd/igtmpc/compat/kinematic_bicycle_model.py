"""shim: `from kinematic_bicycle_model import KinematicBicycleModel` -> igtmpc.models (see compat/README.md)"""
from igtmpc.models import KinematicBicycleModel  # noqa: F401
