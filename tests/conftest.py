import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_quota():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


# the GPU box shows every hardware thread of the host but grants a CPU quota: keep OpenMP (generators, oracle) inside it
os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_quota()))
# the library reads its experiment knobs (GAB_* switches) once per handle; the tests flip them between calls of one handle
# (gab_internal.h: gab_tuning) -- test_bsw_gpu.py::test_knobs_are_read_when_the_handle_is_made covers the shipping behaviour
os.environ.setdefault("GAB_TUNING_LIVE", "1")
# a SIGABRT inside the library or the HIP runtime prints its native call stack (gab_core.hip): one of ~10 full GPU runs of r04 ended in a
# bare "Fatal Python error: Aborted" inside gab_fmi_seed that 24 repetitions of the fmi tests and three more full runs did not show again
os.environ.setdefault("GAB_ABORT_TRACE", os.path.join(ROOT, "gpurun_out", "abort_trace.log"))      # (a file: pytest captures fd 2)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
# What those aborts were (the trace of the third: libhsa-runtime64's event thread -> abort, i.e. a GPU memory fault): by default
# the HIP runtime PINS pageable host memory in place for copies of more than 1 MiB ("HSA Copy Using Pinned resource", rocblit.cpp) and
# keeps the last few pins cached per queue.  A test process frees such an array and gets the next one of the same size at the SAME
# address: the cached pin -- or a gab_host_register of the new array, which the driver layer merges with the registration it still
# has for that address -- then refers to pages that are gone; if the kernel driver's re-validation runs in the window between the
# unmap and the new mapping, the GPU mapping stays invalid and the next copy or written-through store faults.  Staging every
# pageable copy (a very large minimum size for pinned transfers, MiB) leaves no such pins behind -- set before the runtime starts.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1000000")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """build the generators, the CPU oracle and (in the build container) libgab_hip.so once per session"""
    from tools import gabgen
    from oracle import pyoracle
    gabgen.build()
    pyoracle.build(with_ref=False)
    import genarchbench_amd
    so = os.path.join(ROOT, "genarchbench_amd", "libgab_hip.so")
    if not os.path.exists(so):
        genarchbench_amd.build()
    yield
