"""radx_amd — MI355X (gfx950) LSD radix sort behind the RadX host API.

The engine is hand-written HIP in radx_amd/csrc/, exported through the C ABI of
include/radx_hip.h and wrapped for C++ callers by the header-only layer in include/radx/.
This package is the Python driver the tests and bench.py use to reach that C ABI.
"""
from .capi import LIB_PATH, RadxError, lib  # noqa: F401  (import loads the .so or raises)
from .sorter import (Sorter, check_sorted, checksum, generate_u32, generate_u64,  # noqa: F401
                     indiction, permutation, sort_noalloc)

__all__ = ["Sorter", "RadxError", "generate_u32", "generate_u64", "check_sorted", "checksum", "indiction", "permutation", "sort_noalloc", "lib", "LIB_PATH"]
