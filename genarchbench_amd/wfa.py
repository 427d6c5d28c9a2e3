"""Host-side mirror of affine_wavefronts_align + CIGAR copy (wfa/tools/align_benchmark.c:415-437)."""
import ctypes as C

import numpy as np

from ._lib import check, lib


class GabWfaPenalties(C.Structure):
    _fields_ = [("mismatch", C.c_int32), ("gap_opening", C.c_int32), ("gap_extension", C.c_int32)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ops_layout(batch):
    """byte offsets giving every pair pattern_length + text_length bytes of room (edit_cigar_allocate)"""
    cap = batch.pat_len.astype(np.int64) + batch.txt_len.astype(np.int64)
    off = np.zeros(batch.n, np.int64)
    if batch.n > 1:
        np.cumsum(cap[:-1], out=off[1:])
    return off, int(cap.sum())


class AffineWavefronts:
    """affine_wavefronts_new_complete, or -- with min_wavefront_length >= 0 -- affine_wavefronts_new_reduced
    (wfa/gap_affine/affine_wavefront.c:141-181): the adaptive wavefront reduction"""

    def __init__(self, mismatch=4, gap_opening=6, gap_extension=2, device=0, min_wavefront_length=-1, max_distance_threshold=-1):
        p = GabWfaPenalties(mismatch, gap_opening, gap_extension)
        self._h = C.c_void_p()
        check(lib().gab_wfa_create_reduced(C.byref(p), C.c_int(min_wavefront_length), C.c_int(max_distance_threshold),
                                           C.c_int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().gab_wfa_destroy(self._h)
            self._h = None

    __del__ = close

    def align(self, batch):
        """PairBatch -> (ops slab uint8, ops_off, ops_len, score)"""
        off, total = ops_layout(batch)
        ops = np.zeros(total + 16, np.uint8)
        ln = np.full(batch.n, -1, np.int32); sc = np.full(batch.n, -1, np.int32)
        check(lib().gab_wfa_run(self._h, _p(batch.pat), _p(batch.pat_off), _p(batch.pat_len), _p(batch.txt),
                                _p(batch.txt_off), _p(batch.txt_len), C.c_int64(batch.n), _p(ops), _p(off), _p(ln), _p(sc)))
        return ops, off, ln, sc

    def align_packed(self, batch, capacity=None):
        """PairBatch -> (text uint8, text_off, text_len, score): per pair the run-length CIGAR text edit_cigar_print writes
        (gab_wfa_run_packed).  capacity=None sizes the text buffer at a quarter of the operation room and retries once
        with the exact size when the library says it does not fit (GAB_ERANGE)."""
        from ._lib import GabError
        cap = int(capacity) if capacity is not None else int(ops_layout(batch)[1] // 4 + 4096)
        off = np.full(batch.n, -1, np.int64); ln = np.full(batch.n, -1, np.int32); sc = np.full(batch.n, -1, np.int32)
        need = C.c_int64(0)
        for _ in range(2):
            text = np.zeros(cap + 16, np.uint8)
            rc = lib().gab_wfa_run_packed(self._h, _p(batch.pat), _p(batch.pat_off), _p(batch.pat_len), _p(batch.txt), _p(batch.txt_off),
                                          _p(batch.txt_len), C.c_int64(batch.n), _p(text), C.c_int64(cap), _p(off), _p(ln), _p(sc), C.byref(need))
            if rc == -34 and capacity is None and need.value > cap:
                cap = need.value
                continue
            check(rc)
            return text[:need.value], off, ln, sc
        raise GabError(-34, "gab_wfa_run_packed: text does not fit")

    def run_device(self, pat, pat_off, pat_len, txt, txt_off, txt_len, ops, ops_off, ops_len, score, stream=0):
        n = pat_len.numel()
        check(lib().gab_wfa_run_device(self._h, C.c_void_p(pat.data_ptr()), C.c_int64(pat.numel()),
                                       C.c_void_p(pat_off.data_ptr()), C.c_void_p(pat_len.data_ptr()),
                                       C.c_void_p(txt.data_ptr()), C.c_int64(txt.numel()),
                                       C.c_void_p(txt_off.data_ptr()), C.c_void_p(txt_len.data_ptr()), C.c_int64(n),
                                       C.c_void_p(ops.data_ptr()), C.c_void_p(ops_off.data_ptr()),
                                       C.c_void_p(ops_len.data_ptr()), C.c_void_p(score.data_ptr()), C.c_void_p(stream)))

    def run_packed_device(self, pat, pat_off, pat_len, txt, txt_off, txt_len, ops, ops_off, capacity):
        """device tensors in (as run_device; `ops` / `ops_off` = operation room on the device), the printed text out to host
        arrays: (text uint8, text_off, text_len, score, bytes needed) -- gab_wfa_run_packed_device.  Raises GabError(-34) when
        `capacity` bytes do not hold the text."""
        n = pat_len.numel()
        text = np.zeros(int(capacity) + 16, np.uint8)
        off = np.full(n, -1, np.int64); ln = np.full(n, -1, np.int32); sc = np.full(n, -1, np.int32)
        need = C.c_int64(0)
        check(lib().gab_wfa_run_packed_device(self._h, C.c_void_p(pat.data_ptr()), C.c_int64(pat.numel()), C.c_void_p(pat_off.data_ptr()),
                                              C.c_void_p(pat_len.data_ptr()), C.c_void_p(txt.data_ptr()), C.c_int64(txt.numel()),
                                              C.c_void_p(txt_off.data_ptr()), C.c_void_p(txt_len.data_ptr()), C.c_int64(n),
                                              C.c_void_p(ops.data_ptr()), C.c_void_p(ops_off.data_ptr()), _p(text), C.c_int64(int(capacity)),
                                              _p(off), _p(ln), _p(sc), C.byref(need)))
        return text[:need.value], off, ln, sc, need.value

    def last_stats(self):
        w = C.c_int64(0); r = C.c_int64(0); a = C.c_float(0); b = C.c_float(0)
        check(lib().gab_wfa_last_stats(self._h, C.byref(w), C.byref(r), C.byref(a), C.byref(b)))
        return {"work": w.value, "requeued": r.value, "kernel_ms": a.value, "total_ms": b.value}
