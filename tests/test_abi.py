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
