"""shim: `from VehicleReference import VehicleReference` -> igtmpc.vehicle (see compat/README.md)"""
from igtmpc.vehicle import VehicleReference  # noqa: F401
