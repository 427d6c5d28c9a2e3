"""Host-side mirror of benchmark_edit_bpm (bpm/benchmark/benchmark_edit.c:31-56) over the C ABI."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class BpmEngine:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().gab_bpm_create(C.c_int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().gab_bpm_destroy(self._h)
            self._h = None

    __del__ = close

    def benchmark_edit_bpm(self, batch):
        """batch: PairBatch with the driver's longer-is-pattern swap applied -> printed scores (int32, <= 0)"""
        out = np.full(batch.n, 12345, np.int32)
        check(lib().gab_bpm_run(self._h, _p(batch.pat), _p(batch.pat_off), _p(batch.pat_len), _p(batch.txt),
                                _p(batch.txt_off), _p(batch.txt_len), C.c_int64(batch.n), _p(out)))
        return out

    def run_device(self, pat, pat_off, pat_len, txt, txt_off, txt_len, score, stream=0):
        n = pat_len.numel()
        check(lib().gab_bpm_run_device(self._h, C.c_void_p(pat.data_ptr()), C.c_int64(pat.numel()),
                                       C.c_void_p(pat_off.data_ptr()), C.c_void_p(pat_len.data_ptr()),
                                       C.c_void_p(txt.data_ptr()), C.c_int64(txt.numel()),
                                       C.c_void_p(txt_off.data_ptr()), C.c_void_p(txt_len.data_ptr()),
                                       C.c_int64(n), C.c_void_p(score.data_ptr()), C.c_void_p(stream)))

    def last_stats(self):
        st = C.c_int64(0); fp = C.c_int64(0); k = C.c_float(0); t = C.c_float(0)
        check(lib().gab_bpm_last_stats(self._h, C.byref(st), C.byref(fp), C.byref(k), C.byref(t)))
        return {"block_steps": st.value, "full_pairs": fp.value, "kernel_ms": k.value, "total_ms": t.value}
