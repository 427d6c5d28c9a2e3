"""Host-side mirror of the reference's bsw interface (bsw/src/bandedSWA.h:163-245) over the C ABI.

    sw = BandedPairWiseSW(o_del, e_del, o_ins, e_ins, zdrop, end_bonus, mat)   # ctor args as the reference
    scores = sw.getScores16(batch)                                               # one call over all pairs
"""
import ctypes as C

import numpy as np

from ._lib import check, lib


class GabBswParams(C.Structure):
    _fields_ = [("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32), ("e_ins", C.c_int32),
                ("zdrop", C.c_int32), ("end_bonus", C.c_int32), ("w", C.c_int32), ("mat", C.c_int8 * 25)]


RESULT_FIELDS = ("score", "qle", "tle", "gtle", "gscore", "max_off")


def bwa_fill_scmat(a, b, ambig):
    """5x5 score matrix of the reference driver (bsw/src/main_banded.cpp:94-102)"""
    m = np.full((5, 5), -b, np.int8)
    np.fill_diagonal(m, a)
    m[4, :] = ambig
    m[:, 4] = ambig
    return m.reshape(25)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class BandedPairWiseSW:
    def __init__(self, o_del=6, e_del=1, o_ins=6, e_ins=1, zdrop=100, end_bonus=5, mat=None, w=100, device=0):
        mat = bwa_fill_scmat(1, 4, -1) if mat is None else np.asarray(mat, np.int8).reshape(25)
        p = GabBswParams(o_del, e_del, o_ins, e_ins, zdrop, end_bonus, w)
        for i in range(25):
            p.mat[i] = int(mat[i])
        self._h = C.c_void_p()
        check(lib().gab_bsw_create(C.byref(p), C.c_int(device), C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().gab_bsw_destroy(self._h)
            self._h = None

    __del__ = close

    def getScores16(self, batch):
        """host buffers in, int32 scores out (SeqPair.score of every pair, in input order)"""
        out = np.full(batch.n, -12345, np.int32)
        check(lib().gab_bsw_run(self._h, _p(batch.ref), _p(batch.ref_off), _p(batch.qry), _p(batch.qry_off),
                                _p(batch.len1), _p(batch.len2), _p(batch.h0), C.c_int64(batch.n), _p(out)))
        return out

    def run_device(self, ref, ref_off, qry, qry_off, len1, len2, h0, score_out, result_out=None, stream=0):
        """torch CUDA tensors (uint8/int64/int32), asynchronous on `stream` (a raw hipStream_t value)"""
        n = len1.numel()
        check(lib().gab_bsw_run_device(
            self._h, C.c_void_p(ref.data_ptr()), C.c_int64(ref.numel()), C.c_void_p(ref_off.data_ptr()),
            C.c_void_p(qry.data_ptr()), C.c_int64(qry.numel()), C.c_void_p(qry_off.data_ptr()),
            C.c_void_p(len1.data_ptr()), C.c_void_p(len2.data_ptr()), C.c_void_p(h0.data_ptr()), C.c_int64(n),
            C.c_void_p(score_out.data_ptr()),
            C.c_void_p(result_out.data_ptr()) if result_out is not None else C.c_void_p(0),
            C.c_void_p(stream)))

    def last_stats(self):
        cells = C.c_int64(0); kms = C.c_float(0); tms = C.c_float(0)
        check(lib().gab_bsw_last_stats(self._h, C.byref(cells), C.byref(kms), C.byref(tms)))
        return {"cells": cells.value, "kernel_ms": kms.value, "total_ms": tms.value}
