"""MI355X-native batched relative-pose EKF engine (drop-in for the hot path of
mbrymer/quadrotor_landing's quad_state_estimation filter core)."""
from ._lib import LIB_PATH, QLE_F32, QLE_F64, QleDerived, QleError, QleParams, QleSynthCfg, lib  # noqa: F401
from .ekf import BatchedRelativePoseEKF, InputSequence  # noqa: F401
from .twin import RelativePoseEKF  # noqa: F401
from .params import default_params, derive, load_yaml, make_params, set_fields  # noqa: F401
from . import replay  # noqa: F401
