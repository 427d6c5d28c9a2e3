"""Loader for libgab_hip.so.  Fails loudly when the library is missing: there is no CPU fallback."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("GAB_LIB_PATH") or os.path.join(_HERE, "libgab_hip.so")   # override: A/B builds while tuning
_lib = None


class GabError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgab_hip error {code}: {msg}")
        self.code = code


def build(force=False):
    """compile every HIP translation unit for gfx950 into genarchbench_amd/libgab_hip.so (in-tree)"""
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-C", csrc, "-s"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(
                f"{_SO} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C genarchbench_amd/csrc).  There is no CPU fallback.")
        # PyTorch wheels bundle their own libamdhip64.so.7; a process must hold ONE HIP runtime.  When torch is
        # installed, load it first so that libgab_hip.so binds to the runtime torch uses (same soname) -- otherwise
        # a later `torch.cuda` initialisation in the same process finds no GPU.  The C drivers never see torch.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(_SO)
        _lib.gab_version.restype = C.c_char_p
        _lib.gab_last_error.restype = C.c_char_p
    return _lib


def version():
    return lib().gab_version().decode()


def check(rc):
    if rc != 0:
        raise GabError(rc, lib().gab_last_error().decode())
