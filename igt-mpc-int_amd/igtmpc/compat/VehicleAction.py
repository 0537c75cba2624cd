"""shim: `from VehicleAction import VehicleAction` -> igtmpc.vehicle (see compat/README.md)"""
from igtmpc.vehicle import VehicleAction  # noqa: F401
