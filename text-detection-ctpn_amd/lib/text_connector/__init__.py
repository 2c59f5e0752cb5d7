from .detectors import TextDetector  # noqa: F401
from .text_connect_cfg import Config  # noqa: F401
