"""genarchbench_amd -- MI355X (gfx950) engine for the GenArchBench banded-DP / seed-chaining hot path.

The product is libgab_hip.so (hand-written HIP kernels behind the C ABI in include/gab.h) and the
C drivers under benchmarks/.  This Python package is only the thin ctypes mirror used by the tests
and bench.py; it never falls back to a CPU implementation.
"""
from ._lib import GabError, build, lib, version  # noqa: F401
