"""Host-side mirror of benchmark_bitpal_m0_x1_g1 / benchmark_bitpal_m1_x4_g2 (bpm/benchmark/benchmark_bitpal.c:30-54)
over the C ABI: the bpm driver's `-a bitpal-edit` / `-a bitpal-scored`."""
import ctypes as C

import numpy as np

from ._lib import check, lib

BITPAL_EDIT, BITPAL_SCORED = 0, 1


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class BitpalEngine:
    def __init__(self, algorithm, device=0):
        self._h = C.c_void_p()
        check(lib().gab_bitpal_create(C.c_int(algorithm), C.c_int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().gab_bitpal_destroy(self._h)
            self._h = None

    __del__ = close

    def benchmark_bitpal(self, batch):
        """batch: PairBatch (swapped or not: the score is symmetric) -> printed scores (int32)"""
        out = np.full(batch.n, 12345, np.int32)
        check(lib().gab_bitpal_run(self._h, _p(batch.pat), _p(batch.pat_off), _p(batch.pat_len), _p(batch.txt),
                                   _p(batch.txt_off), _p(batch.txt_len), C.c_int64(batch.n), _p(out)))
        return out

    def run_device(self, pat, pat_off, pat_len, txt, txt_off, txt_len, score, stream=0):
        n = pat_len.numel()
        check(lib().gab_bitpal_run_device(self._h, C.c_void_p(pat.data_ptr()), C.c_int64(pat.numel()),
                                          C.c_void_p(pat_off.data_ptr()), C.c_void_p(pat_len.data_ptr()),
                                          C.c_void_p(txt.data_ptr()), C.c_int64(txt.numel()),
                                          C.c_void_p(txt_off.data_ptr()), C.c_void_p(txt_len.data_ptr()),
                                          C.c_int64(n), C.c_void_p(score.data_ptr()), C.c_void_p(stream)))

    def last_stats(self):
        ce = C.c_int64(0); lp = C.c_int64(0); k = C.c_float(0); t = C.c_float(0)
        check(lib().gab_bitpal_last_stats(self._h, C.byref(ce), C.byref(lp), C.byref(k), C.byref(t)))
        return {"cells": ce.value, "long_pairs": lp.value, "kernel_ms": k.value, "total_ms": t.value}
