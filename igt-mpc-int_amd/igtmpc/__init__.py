"""igtmpc -- MI355X-native batched MPC rollout + shooting solver.

Drop-in for the per-timestep optimisation inner loop of hansungkim98122/IGT-MPC-INT
(mpc.py over common/kinematic_bicycle_model_frenet.py): HIP kernels behind the C ABI in
include/igtmpc.h, bound here with ctypes.  See DESIGN.md."""
from ._lib import IgtError, load as load_library  # noqa: F401
from .solver import BatchSolver  # noqa: F401
from .vehicle import Curvature, VehicleAction, VehicleReference, VehicleState  # noqa: F401
from .predictor import ConstantAccelerationModel, PredictorBase  # noqa: F401
from .models import KinematicBicycleModel, KinematicBicycleModelFrenet  # noqa: F401
from .planner import MPC_Planner  # noqa: F401
from .value_nets import shipped_value_net  # noqa: F401

__all__ = ['BatchSolver', 'IgtError', 'load_library', 'MPC_Planner', 'VehicleState', 'VehicleAction',
           'VehicleReference', 'Curvature', 'PredictorBase', 'ConstantAccelerationModel',
           'KinematicBicycleModelFrenet', 'KinematicBicycleModel']
