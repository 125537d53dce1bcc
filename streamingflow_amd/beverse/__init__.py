"""BEVerse-named counterparts (mmdet3d/models/beverse/models/{basic,motion}_modules.py of the
reference tree): ``FuturePrediction``, ``SpatialDistributionModule``, ``DistributionModule`` — the
class names BASELINE.json's north star lists.  Secondary, signature-compatible:
the shipped StreamingFlow evaluation never reaches them."""
from .motion_modules import DistributionEncoder, DistributionModule, FuturePrediction, SpatialDistributionModule  # noqa: F401
from .basic_modules import Bottleneck, ConvBlock, SpatialGRU  # noqa: F401
