"""shim: `from PredictorBase import PredictorBase` -> igtmpc.predictor (see compat/README.md)"""
from igtmpc.predictor import PredictorBase  # noqa: F401
