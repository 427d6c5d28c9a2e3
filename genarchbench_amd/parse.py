"""Host-side mirror of the GPU input parsers (include/gab.h, "input parsers"): the whole text file in, packed device
buffers out.  Used by the tests and bench.py; the C drivers call the same entry points (GAB_GPU_PARSE=1)."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class BswPacked(C.Structure):
    _fields_ = [("n", C.c_int64), ("d_ref", C.c_void_p), ("d_ref_off", C.c_void_p), ("d_qry", C.c_void_p),
                ("d_qry_off", C.c_void_p), ("d_len1", C.c_void_p), ("d_len2", C.c_void_p), ("d_h0", C.c_void_p),
                ("ref_bytes", C.c_int64), ("qry_bytes", C.c_int64)]


class PairsPacked(C.Structure):
    _fields_ = [("n", C.c_int64), ("d_text", C.c_void_p), ("text_bytes", C.c_int64), ("d_pat_off", C.c_void_p), ("d_txt_off", C.c_void_p),
                ("d_pat_len", C.c_void_p), ("d_txt_len", C.c_void_p), ("d_cap_off", C.c_void_p), ("cap_bytes", C.c_int64)]


class ChainHdr(C.Structure):
    _fields_ = [("n", C.c_int64), ("avg_qspan", C.c_float), ("max_dist_x", C.c_int32), ("max_dist_y", C.c_int32),
                ("bw", C.c_int32), ("n_segs", C.c_int32)]


class ChainPacked(C.Structure):
    _fields_ = [("ncalls", C.c_int64), ("total", C.c_int64), ("d_x", C.c_void_p), ("d_y", C.c_void_p),
                ("call_off", C.POINTER(C.c_int64)), ("hdr", C.c_void_p)]


def _d2h(ptr, nbytes, dtype):
    out = np.zeros(max(nbytes // np.dtype(dtype).itemsize, 1), dtype)
    if nbytes:
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), C.c_int(2)) == 0
    return out[:nbytes // np.dtype(dtype).itemsize]


class InputParser:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().gab_parser_create(C.c_int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().gab_parser_destroy(self._h)
            self._h = None

    __del__ = close

    def bsw_pairs(self, text: bytes, stream=0):
        """bsw input text (main_banded.cpp:152-206) -> BswPacked (device pointers owned by the parser)"""
        out = BswPacked()
        buf = np.frombuffer(text, np.uint8)
        check(lib().gab_bsw_parse_pairs(self._h, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)), C.byref(out), C.c_void_p(stream)))
        return out

    def bsw_pairs_device(self, d_text, nbytes, stream=0):
        out = BswPacked()
        check(lib().gab_bsw_parse_pairs_device(self._h, C.c_void_p(d_text), C.c_int64(nbytes), C.byref(out), C.c_void_p(stream)))
        return out

    def pairs(self, text: bytes, swap_longer_first, stream=0):
        """bpm / wfa input text ('>' / '<' line pairs) -> PairsPacked"""
        out = PairsPacked()
        buf = np.frombuffer(text, np.uint8)
        check(lib().gab_pairs_parse(self._h, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)), C.c_int(1 if swap_longer_first else 0),
                                    C.byref(out), C.c_void_p(stream)))
        return out

    def chain(self, text: bytes, stream=0):
        """chain / fast-chain input text -> ChainPacked (anchors on the device, call table on the host)"""
        out = ChainPacked()
        buf = np.frombuffer(text, np.uint8)
        check(lib().gab_chain_parse(self._h, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)), C.byref(out), C.c_void_p(stream)))
        return out

    @staticmethod
    def chain_to_host(pk):
        from tools.gabgen import CHAIN_HDR
        n = pk.ncalls
        off = np.ctypeslib.as_array(pk.call_off, shape=(n + 1,)).copy()
        hdr = np.ctypeslib.as_array(C.cast(pk.hdr, C.POINTER(C.c_uint8)), shape=(n * CHAIN_HDR.itemsize,)).copy().view(CHAIN_HDR)
        return {"call_off": off, "hdr": hdr, "x": _d2h(pk.d_x, 8 * pk.total, np.uint64), "y": _d2h(pk.d_y, 8 * pk.total, np.uint64)}

    def last_stats(self):
        ms = C.c_float(0)
        check(lib().gab_parser_last_stats(self._h, C.byref(ms)))
        return {"kernel_ms": ms.value}

    # ---- copies for tests
    @staticmethod
    def bsw_to_host(pk):
        n = pk.n
        len1 = _d2h(pk.d_len1, 4 * n, np.int32); len2 = _d2h(pk.d_len2, 4 * n, np.int32)
        roff = _d2h(pk.d_ref_off, 8 * n, np.int64); qoff = _d2h(pk.d_qry_off, 8 * n, np.int64)
        tot1 = int(roff[-1] + len1[-1]) if n else 0; tot2 = int(qoff[-1] + len2[-1]) if n else 0
        return {"len1": len1, "len2": len2, "h0": _d2h(pk.d_h0, 4 * n, np.int32), "ref_off": roff, "qry_off": qoff,
                "ref": _d2h(pk.d_ref, tot1, np.uint8), "qry": _d2h(pk.d_qry, tot2, np.uint8)}

    @staticmethod
    def pairs_to_host(pk):
        n = pk.n
        return {"pat_off": _d2h(pk.d_pat_off, 8 * n, np.int64), "txt_off": _d2h(pk.d_txt_off, 8 * n, np.int64),
                "pat_len": _d2h(pk.d_pat_len, 4 * n, np.int32), "txt_len": _d2h(pk.d_txt_len, 4 * n, np.int32),
                "cap_off": _d2h(pk.d_cap_off, 8 * (n + 1), np.int64)}
