"""shim: `from VehicleState import VehicleState` -> igtmpc.vehicle (see compat/README.md)"""
from igtmpc.vehicle import VehicleState  # noqa: F401
