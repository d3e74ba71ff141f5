from ..detector import TemporalDetector  # noqa: F401
