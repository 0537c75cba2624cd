"""shim: `from mpc import MPC_Planner` -> igtmpc.planner (see compat/README.md)"""
from igtmpc.planner import MPC_Planner  # noqa: F401
