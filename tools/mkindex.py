"""ctypes front-end of tools/mkindex (FM-index builder in the BWA-MEM2 .bwt.2bit.64 layout)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mkindex")
_lib = None


class GabFmIndex(C.Structure):
    _fields_ = [("ref_seq_len", C.c_int64), ("count", C.c_int64 * 5), ("cp_occ_size", C.c_int64),
                ("cp_occ", C.c_void_p), ("sentinel_index", C.c_int64), ("n_sa", C.c_int64),
                ("sa_ms_byte", C.c_void_p), ("sa_ls_word", C.c_void_p)]


def build_tools():
    so = os.path.join(_DIR, "libgabmkindex.so")
    if not os.path.exists(so) or not os.path.exists(os.path.join(_DIR, "gab-mkindex")):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_tools())
    return _lib


class FmIndex:
    """in-memory index: ref_seq_len, count[5] (file convention, not yet +1), cp_occ bytes, sentinel_index"""

    def __init__(self, fwd_codes, power=0):
        """power = m > 0: the index of the reference (U . revcomp(U))^m with U = fwd_codes, produced without sorting the
        whole text (gab_mkindex_build_power): how the tests get an index of >= 2^32 rows in seconds"""
        fwd = np.ascontiguousarray(fwd_codes, np.uint8)
        self._fwd = fwd
        self._raw = GabFmIndex()
        if power:
            rc = lib().gab_mkindex_build_power(fwd.ctypes.data_as(C.c_void_p), C.c_int64(len(fwd)), C.c_int64(power), C.byref(self._raw))
        else:
            rc = lib().gab_mkindex_build(fwd.ctypes.data_as(C.c_void_p), C.c_int64(len(fwd)), C.byref(self._raw))
        if rc:
            raise RuntimeError(f"gab_mkindex_build{'_power' if power else ''} failed ({rc})")
        r = self._raw
        self.ref_seq_len = r.ref_seq_len
        self.count = np.array(list(r.count), np.int64)
        self.sentinel_index = r.sentinel_index
        self.cp_occ = np.ctypeslib.as_array(C.cast(r.cp_occ, C.POINTER(C.c_uint8)), shape=(r.cp_occ_size * 64,))
        # sampled suffix array (one entry per 8 rows): most-significant bytes and low words, FMI_search.cpp:439-447
        self.sa_ms_byte = np.ctypeslib.as_array(C.cast(r.sa_ms_byte, C.POINTER(C.c_int8)), shape=(r.n_sa,))
        self.sa_ls_word = np.ctypeslib.as_array(C.cast(r.sa_ls_word, C.POINTER(C.c_uint32)), shape=(r.n_sa,))

    def write(self, prefix, with_bns=False, name="synthetic"):
        """<prefix>.bwt.2bit.64; with_bns also writes the .ann/.amb/.pac files the reference's loader needs"""
        if lib().gab_mkindex_write(C.byref(self._raw), prefix.encode()):
            raise IOError(f"cannot write {prefix}.bwt.2bit.64")
        if with_bns:
            if lib().gab_mkindex_write_bns(prefix.encode(), self._fwd.ctypes.data_as(C.c_void_p), C.c_int64(len(self._fwd)),
                                           name.encode()):
                raise IOError(f"cannot write {prefix}.ann/.amb/.pac")

    def close(self):
        if self._raw.cp_occ:
            lib().gab_mkindex_free(C.byref(self._raw))
            self.cp_occ = None; self.sa_ms_byte = None; self.sa_ls_word = None

    __del__ = close


class IndexFile:
    """the arrays of an index file <prefix>.bwt.2bit.64 as the reference writes it (FMI_search.cpp:163-164,252,275-276,296:
    i64 reference_seq_len, i64 count[5], CP_OCC[(len >> 6) + 1], the sampled suffix array as (len >> 3) + 1 most-significant
    bytes and as many low words, i64 sentinel_index), memory-mapped: same attributes as FmIndex.  bench.py's ranks share ONE
    index built by rank 0 this way."""

    def __init__(self, prefix):
        path = prefix + ".bwt.2bit.64"
        mm = np.memmap(path, np.uint8, "r")
        head = np.frombuffer(mm[:48].tobytes(), np.int64)
        self.ref_seq_len = int(head[0])
        self.count = head[1:6].copy()
        nocc = (self.ref_seq_len >> 6) + 1
        nsa = (self.ref_seq_len >> 3) + 1
        o = 48
        self.cp_occ = mm[o:o + 64 * nocc]; o += 64 * nocc
        self.sa_ms_byte = mm[o:o + nsa].view(np.int8); o += nsa
        self.sa_ls_word = np.frombuffer(mm[o:o + 4 * nsa].tobytes(), np.uint32); o += 4 * nsa
        self.sentinel_index = int(np.frombuffer(mm[o:o + 8].tobytes(), np.int64)[0])
        assert o + 8 == len(mm), f"{path}: unexpected size"
        self._mm = mm

    def close(self):
        self.cp_occ = self.sa_ms_byte = self.sa_ls_word = self._mm = None
