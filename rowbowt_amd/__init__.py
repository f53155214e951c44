"""rowbowt_amd -- MI355X-native batched backward search behind rowbowt's RowBowt API.

The product is the C-ABI shared library rowbowt_amd/librbg.so (include/rbg.h) and the C++17 header
shim rowbowt_amd/include/rowbowt_gpu.hpp.  This Python package is only the ctypes binding the
tests and bench.py drive it through; there is no CPU compute path anywhere in it.
"""
from .capi import (  # noqa: F401
    LoadRbwtFlag,
    RbgError,
    RowBowt,
    lib,
    load_rowbowt,
    pack_reads,
    set_default_option,
)

__all__ = ["LoadRbwtFlag", "RbgError", "RowBowt", "lib", "load_rowbowt", "pack_reads", "set_default_option"]
