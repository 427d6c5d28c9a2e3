"""CPU: libgab_hip.so loads and exports every symbol include/gab.h declares; entry points fail loudly
(no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

import genarchbench_amd
from genarchbench_amd import _lib
from tests.util import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gab.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gab_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol():
    lib = genarchbench_amd.lib()
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f"libgab_hip.so does not export {s}"


def test_version():
    assert "gfx950" in genarchbench_amd.version()


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    from genarchbench_amd.bsw import BandedPairWiseSW
    with pytest.raises(_lib.GabError) as e:
        BandedPairWiseSW()
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    """the product path must not route through oracle/ (see oracle/oracle.h)"""
    bad = []
    for base in ("genarchbench_amd", "benchmarks", "include", "tools"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".c", ".h", ".cpp", ".hip", ".sh")) or fn == "Makefile":
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"pyoracle|liboracle|oracle/|oracle\.h", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_pageable_copies_are_staged_by_default():
    """importing the package sets the HIP runtime's minimum size for pinned transfers (a default: the caller's value wins) before the
    runtime starts -- pageable numpy arrays are then staged, not pinned in place (genarchbench_amd/__init__.py, DESIGN.md lesson 16)"""
    import subprocess
    import sys
    code = "import os; os.environ.pop('GPU_PINNED_MIN_XFER_SIZE', None); import genarchbench_amd; print(os.environ['GPU_PINNED_MIN_XFER_SIZE'])"
    assert subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, text=True).strip() == "1000000"
    code = "import os; os.environ['GPU_PINNED_MIN_XFER_SIZE'] = '64'; import genarchbench_amd; print(os.environ['GPU_PINNED_MIN_XFER_SIZE'])"
    assert subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, text=True).strip() == "64"


def test_abort_trace_keeps_the_stack_and_the_end_of_a_captured_stderr(tmp_path):
    """GAB_ABORT_TRACE=<file>: a SIGABRT anywhere in the process appends its native stack to the file and, when file descriptor 2 is
    a regular file (a test runner's capture), the end of that file too -- the HSA runtime's fault message would otherwise die with
    the process (gab_core.hip)"""
    import signal
    import subprocess
    import sys
    trace = tmp_path / "trace.log"
    code = ("import os, ctypes, tempfile\n"
            "f = tempfile.TemporaryFile(); os.dup2(f.fileno(), 2)\n"
            "os.write(2, b'Memory access fault by GPU node-9 (a line written by the test)\\n')\n"
            "import genarchbench_amd; genarchbench_amd.lib()\n"
            "ctypes.CDLL(None).abort()\n")
    env = dict(os.environ, GAB_ABORT_TRACE=str(trace))
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env)
    assert p.returncode == -signal.SIGABRT
    text = trace.read_text()
    assert "[gab] SIGABRT -- native stack:" in text and "abort" in text
    assert "Memory access fault by GPU node-9 (a line written by the test)" in text


def test_every_copy_of_the_library_goes_through_its_wrappers():
    """gab_internal.h turns hipMemcpy / hipMemcpyAsync of the library's translation units into gab_memcpy / gab_memcpy_async (pageable
    memory staged by the library under GAB_STAGE_PAGEABLE=1); only gab_core.hip, which defines them, may opt out"""
    import glob
    csrc = os.path.join(ROOT, "genarchbench_amd", "csrc")
    hdr = open(os.path.join(csrc, "gab_internal.h")).read()
    assert "#define hipMemcpyAsync gab_memcpy_async" in hdr and "#define hipMemcpy gab_memcpy" in hdr
    units = sorted(glob.glob(os.path.join(csrc, "*.hip")))
    assert len(units) >= 8
    for u in units:
        src = open(u).read()
        assert '#include "gab_internal.h"' in src or '#include "chain_dev.h"' in src, u
        assert ("GAB_NO_COPY_MACROS" in src) == (os.path.basename(u) == "gab_core.hip"), u
        assert not re.search(r"\bhipMemcpy(2D|3D|Peer|DtoH|HtoD)\w*\s*\(", src), f"{u}: a copy call the wrappers do not cover"
