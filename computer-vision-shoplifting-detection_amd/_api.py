"""Public names of the package (imported as ``cvsd_amd``)."""
from .graph import build_program, parse_model_name  # noqa: F401

__all__ = ["build_program", "parse_model_name"]
