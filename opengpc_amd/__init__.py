"""opengpc_amd -- MI355X-native hot path of the openGPC sparse-stereo matcher.

The product is libgpc_hip.so (hand-written HIP for gfx950, C ABI in include/gpc_hip.h) and
the C++ host API in include/gpc/ that mirrors the reference's gpc::inference::Forest.
This Python package is plumbing for tests, bench.py and multi-GPU launches: a ctypes
binding (capi) and the in-tree build recipe (build).
"""
from . import build as _build  # noqa: F401
from .capi import (Context, TrainSet, SPLIT_DTYPE, STATS_DTYPE, FilterMask, GpcError, Settings, SUPPORT_DTYPE, CORR_DTYPE, load, parse_forest,
                   read_forest)

__all__ = ["Context", "TrainSet", "SPLIT_DTYPE", "STATS_DTYPE", "FilterMask", "GpcError", "Settings", "SUPPORT_DTYPE", "CORR_DTYPE", "load",
           "parse_forest", "read_forest"]
