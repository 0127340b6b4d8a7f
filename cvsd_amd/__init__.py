"""Importable alias for the ``computer-vision-shoplifting-detection_amd/`` package directory
(its name contains hyphens, so Python cannot import it directly).  This alias package simply
extends its ``__path__`` with that directory; all code lives there."""
import os as _os

__path__.append(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..",
                              "computer-vision-shoplifting-detection_amd"))

from ._api import *  # noqa: F401,F403,E402  (resolved from the hyphenated directory)
