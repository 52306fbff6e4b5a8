"""fluidsolver_amd: MI355X (gfx950) implementation of the fluidsolver particle-solver step.

The compute path is libfluidsolver_amd.so (hand-written HIP kernels behind the C ABI of
include/fluidsolver_amd.h); this package is the Python host side over it.
"""
from . import capi, scenes  # noqa: F401
from .capi import Solver, make_params, FsaError  # noqa: F401

__all__ = ["capi", "scenes", "Solver", "make_params", "FsaError"]
