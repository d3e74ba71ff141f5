"""Import-path compatibility with ``openglottal.models`` (models/__init__.py)."""
from ..detector import TemporalDetector  # noqa: F401
from ..unet import UNet  # noqa: F401
