"""streamingflow_amd — MI355X (gfx950) native GRU-ODE future-state predictor of StreamingFlow.

Host side: Python modules mirroring the reference's nn.Module interface for this one hot path
(``models.future_prediction_ode.FuturePredictionODE`` and the layers below it).  Device side:
``libsfnative.so`` (hand-written HIP, C ABI in include/sfnative.h).  There is no CPU fallback.
"""
from .models.future_prediction_ode import FuturePredictionODE  # noqa: F401
from .layers.temporal_ode_bayes import NNFOwithBayesianJumps, DualGRUODECell, DualGRUCell, GRUObservationCell  # noqa: F401
from .layers.temporal import SpatialGRU  # noqa: F401
from .models.lift_splat import LiftSplat  # noqa: F401
from .bev_pool import bev_pool  # noqa: F401

__version__ = "0.1.0"
from .packing import set_math_mode, math_mode, set_winograd, winograd, set_persistent_flow  # noqa: F401,E402  (opt-in "bf16x3"; the default "fp32" is exact)
