"""hercules_amd -- MI355X-native explicit time-stepping hot path of CMU Hercules.

The product is the C-ABI library built from hercules_amd/csrc (HIP kernels for
gfx950 + C host code).  This package only holds the ctypes bindings the tests,
bench.py and __graft_entry__ use to drive it; there is no Python or CPU
implementation of the solver here.
"""
from . import capi
from .capi import (HQ_VARIANT_AUTO, HQ_VARIANT_PATCH, HQ_VARIANT_SCATTER, HqError, Solver,
                   device_count, load_library)

__all__ = ["Solver", "HqError", "device_count", "load_library", "HQ_VARIANT_AUTO",
           "HQ_VARIANT_SCATTER", "HQ_VARIANT_PATCH"]
