"""MI355X-native FSR 1.0 hot path (EASU + RCAS) behind the reference's operator surface.

The directory name carries a hyphen (fidelityfx-fsr_amd), so import it with
``importlib.import_module("fidelityfx-fsr_amd")`` or through the ``fsr1_amd`` alias module at the
repository root.
"""
from . import _lib, frames  # noqa: F401
from .api import *  # noqa: F401,F403
from .api import FSR_Filter, Pipeline, State, Timer  # noqa: F401
from .shard import frames_for_rank  # noqa: F401

build = _lib.build
load = _lib.load
