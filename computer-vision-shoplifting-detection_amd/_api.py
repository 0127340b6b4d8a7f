"""Public names of the package (imported as ``cvsd_amd``)."""
from .graph import build_program, parse_model_name  # noqa: F401
from .results import Boxes, Keypoints, Results  # noqa: F401
from .engine import YOLO  # noqa: F401

__all__ = ["YOLO", "Results", "Boxes", "Keypoints", "build_program", "parse_model_name"]
