"""genarchbench_amd -- MI355X (gfx950) engine for the GenArchBench banded-DP / seed-chaining hot path.

The product is libgab_hip.so (hand-written HIP kernels behind the C ABI in include/gab.h) and the
C drivers under benchmarks/.  This Python package is only the thin ctypes mirror used by the tests
and bench.py; it never falls back to a CPU implementation.
"""
import os as _os

# Pageable host arrays (numpy) handed to the host-pointer entry points, and torch's own copies of such arrays: by default the HIP
# runtime pins them IN PLACE for copies of more than 1 MiB and keeps the last few pins cached; a process that frees an array and gets
# the next one at the same address can meet a pin whose pages are gone -- a GPU memory fault (lesson 16).  A very
# large minimum size for pinned transfers makes the runtime stage such copies instead.  Read when the runtime starts: set it before.
_os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1000000")

from ._lib import GabError, build, lib, version  # noqa: F401
