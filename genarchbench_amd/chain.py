"""Host-side mirror of host_chain_kernel (chain/src/host_kernel.h:6, fast-chain/src/host_kernel.h:6)."""
import ctypes as C

import numpy as np

from ._lib import check, lib

CHAIN, FASTCHAIN = 0, 1


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ChainEngine:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().gab_chain_create(C.c_int(device), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().gab_chain_destroy(self._h)
            self._h = None

    __del__ = close

    def host_chain_kernel(self, batch, mode=CHAIN, pinned=False):
        """batch: tools.gabgen.ChainBatch-like (hdr records, call_off, x, y) -> (scores, parents).
        pinned: page-lock the four arrays for the call, as the drivers do with their slabs (gab_host_register)"""
        score = np.full(batch.nanchors, -777, np.int32); parent = np.full(batch.nanchors, -777, np.int32)
        locked = []
        if pinned:
            for a in (batch.x, batch.y, score, parent):
                if a.nbytes:
                    check(lib().gab_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)))
                    locked.append(a)
        try:
            check(lib().gab_chain_run(self._h, C.c_int(mode), _p(batch.x), _p(batch.y), _p(batch.call_off),
                                      _p(batch.hdr), C.c_int64(batch.ncalls), _p(score), _p(parent)))
        finally:
            for a in locked:
                lib().gab_host_unregister(C.c_void_p(a.ctypes.data))
        return score, parent

    def run_device(self, mode, x, y, call_off, hdr, score, parent, stream=0):
        """x, y, score, parent: torch CUDA tensors; call_off, hdr: numpy (host) call table"""
        check(lib().gab_chain_run_device(self._h, C.c_int(mode), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                         _p(call_off), _p(hdr), C.c_int64(len(hdr)),
                                         C.c_void_p(score.data_ptr()), C.c_void_p(parent.data_ptr()),
                                         C.c_void_p(stream)))

    def run_device_through(self, mode, x, y, call_off, hdr, score, parent, pinned=True, stream=0):
        """run_device + the results in host arrays as well (gab_chain_run_device_through): returns (score, parent) numpy arrays;
        pinned: page-lock them for the call, so that the DP kernel writes them through while it runs"""
        n = int(score.numel())
        hs = np.full(n, -777, np.int32); hp = np.full(n, -777, np.int32)
        locked = []
        if pinned:
            for a in (hs, hp):
                if a.nbytes:
                    check(lib().gab_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)))
                    locked.append(a)
        try:
            check(lib().gab_chain_run_device_through(self._h, C.c_int(mode), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                     _p(call_off), _p(hdr), C.c_int64(len(hdr)), C.c_void_p(score.data_ptr()),
                                                     C.c_void_p(parent.data_ptr()), _p(hs), _p(hp), C.c_void_p(stream)))
        finally:
            for a in locked:
                lib().gab_host_unregister(C.c_void_p(a.ctypes.data))
        return hs, hp

    def last_stats(self):
        ev = C.c_int64(0); ms = C.c_float(0)
        check(lib().gab_chain_last_stats(self._h, C.byref(ev), C.byref(ms)))
        return {"evals": ev.value, "kernel_ms": ms.value}
