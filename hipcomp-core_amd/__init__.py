"""hipcomp-core_amd -- MI355X-native batched lossless codecs (LZ4, Snappy, Cascaded).

The product is ``lib/libhipcomp.so``: a C-ABI shared library that exports the
reference's low-level batched interface (``hipcompBatched{LZ4,Snappy,Cascaded}*``,
declared in ``include/hipcomp/*.h``) on top of hand-written gfx950 kernels.

This Python package is plumbing only: a ctypes binding of that C ABI
(:mod:`.api`) and helpers that lay chunk lists out in HBM with torch tensors
(:mod:`.batch`).  There is no CPU or PyTorch fallback: importing the package
without the built library raises.

The directory name contains a hyphen, so import it with::

    import importlib
    hc = importlib.import_module("hipcomp-core_amd")
"""
from . import api  # noqa: F401  (loads libhipcomp.so, raises if missing)
from .api import (  # noqa: F401
    HipcompLibrary,
    default_library,
    knobs_library,
    hipcompStatus,
    hipcompType,
    LZ4Opts,
    SnappyOpts,
    CascadedOpts,
)
from . import batch  # noqa: F401

__all__ = [
    "api",
    "batch",
    "HipcompLibrary",
    "default_library",
    "knobs_library",
    "hipcompStatus",
    "hipcompType",
    "LZ4Opts",
    "SnappyOpts",
    "CascadedOpts",
]
