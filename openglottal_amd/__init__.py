"""openglottal_amd — MI355X-native glottal segmentation hot path.

Drop-in names of the reference package (`openglottal/__init__.py:5-20`) for the
per-frame segmentation path; everything heavy runs in ``libopenglottal_hip.so``.
"""

__version__ = "0.1.0"

from ._lib import OpenGlottalHipError  # noqa: F401
from .detector import TemporalDetector  # noqa: F401
from .features import _kinematic_features, extract_features_unet  # noqa: F401
from .unet import UNet  # noqa: F401
from .utils import dice, iou, unet_segment_frame  # noqa: F401

__all__ = ["UNet", "TemporalDetector", "extract_features_unet", "unet_segment_frame", "dice", "iou",
           "OpenGlottalHipError"]
