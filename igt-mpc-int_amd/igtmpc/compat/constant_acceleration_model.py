"""shim: `from constant_acceleration_model import ConstantAccelerationModel` -> igtmpc.predictor (see compat/README.md)"""
from igtmpc.predictor import ConstantAccelerationModel  # noqa: F401
