"""Host-side mirror of FMI_search (load_index + the three seeding passes of fmi/fmi.cpp:288-348)."""
import ctypes as C

import numpy as np

from ._lib import check, lib

SMEM_DTYPE = np.dtype([("rid", np.uint32), ("m", np.uint32), ("n", np.uint32), ("pad", np.uint32),
                       ("k", np.int64), ("l", np.int64), ("s", np.int64)])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class FMI_search:
    def __init__(self, prefix=None, device=0, arrays=None):
        """prefix: path prefix of a BWA-MEM2 index (<prefix>.bwt.2bit.64), or arrays=(ref_seq_len, count[5],
        cp_occ bytes, sentinel_index) for an index built in memory"""
        self._h = C.c_void_p()
        if arrays is not None:
            n, count, occ, sent = arrays
            count = np.ascontiguousarray(count, np.int64)
            check(lib().gab_fmi_create(C.c_int(device), C.c_int64(n), _p(count), _p(occ), C.c_int64(sent), C.byref(self._h)))
        else:
            check(lib().gab_fmi_load(C.c_int(device), prefix.encode(), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib().gab_fmi_destroy(self._h)
            self._h = None

    __del__ = close

    def seed(self, reads, min_seed_len=19):
        """reads: .enc [n, stride] uint8, .len int32 -> (smems structured array sorted by rid/m/n desc, read_off)"""
        out = C.c_void_p(); n = C.c_int64(0)
        check(lib().gab_fmi_seed(self._h, _p(reads.enc), C.c_int32(reads.stride), _p(reads.len), C.c_int64(reads.n),
                                 C.c_int32(min_seed_len), C.byref(out), C.byref(n)))
        cnt = n.value
        arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(cnt, 1) * 40,))[:cnt * 40].copy().view(SMEM_DTYPE)
        lib().gab_fmi_free(out)
        off = np.zeros(reads.n + 1, np.int64)
        np.cumsum(np.bincount(arr["rid"], minlength=reads.n), out=off[1:])
        return arr, off

    def clone(self):
        """a second handle on the same GPU sharing this handle's index (gab_fmi_clone): one per host worker thread"""
        other = FMI_search.__new__(FMI_search)
        other._h = C.c_void_p()
        check(lib().gab_fmi_clone(self._h, C.byref(other._h)))
        return other

    def seed_into(self, enc, length, out, min_seed_len=19):
        """host arrays: enc [n, stride] uint8 (C-contiguous rows), length int32, out = the caller's SMEM_DTYPE array
        (page-lock it for the full link rate) -> number of SMEMs written; GabError(GAB_ERANGE) when out is too small"""
        n = C.c_int64(0)
        rc = lib().gab_fmi_seed_into(self._h, _p(enc), C.c_int32(enc.shape[1]), _p(length), C.c_int64(enc.shape[0]),
                                     C.c_int32(min_seed_len), _p(out), C.c_int64(len(out)), C.byref(n))
        check(rc)
        return n.value

    def reserve(self, max_reads, stride):
        """size the device buffers for host-pointer calls of up to max_reads reads now (gab_fmi_reserve): outside a timed region"""
        check(lib().gab_fmi_reserve(self._h, C.c_int64(max_reads), C.c_int32(stride)))

    def seed_device(self, enc, length, min_seed_len=19, stream=0):
        """torch CUDA tensors enc [n, stride] uint8, length int32 -> (device ptr of gab_smem[], device ptr of
        read_off[], count); the pointers stay valid until the next call"""
        d_out = C.c_void_p(); d_off = C.c_void_p(); n = C.c_int64(0)
        check(lib().gab_fmi_seed_device(self._h, C.c_void_p(enc.data_ptr()), C.c_int32(enc.shape[1]),
                                        C.c_void_p(length.data_ptr()), C.c_int64(enc.shape[0]), C.c_int32(min_seed_len),
                                        C.byref(d_out), C.byref(d_off), C.byref(n), C.c_void_p(stream)))
        return d_out.value, d_off.value, n.value

    def last_stats(self):
        e = C.c_int64(0); n = C.c_int64(0); k = C.c_float(0)
        check(lib().gab_fmi_last_stats(self._h, C.byref(e), C.byref(n), C.byref(k)))
        r = C.c_int64(0)
        check(lib().gab_fmi_last_records(self._h, C.byref(r)))
        return {"ext_calls": e.value, "smems": n.value, "kernel_ms": k.value, "cp_occ_records": r.value}

    # ---- suffix-array look-up (FMI_search::get_sa_entries, FMI_search.cpp:1177-1196)
    def set_sa(self, sa_ms_byte, sa_ls_word):
        ms = np.ascontiguousarray(sa_ms_byte, np.int8); ls = np.ascontiguousarray(sa_ls_word, np.uint32)
        check(lib().gab_fmi_set_sa(self._h, _p(ms), _p(ls)))

    def get_sa_entries(self, smems, max_occ):
        """smems: structured array (SMEM_DTYPE) -> (coords int64[total], coord_off int64[n+1])"""
        sm = np.ascontiguousarray(smems)
        co = C.c_void_p(); off = C.c_void_p(); tot = C.c_int64(0)
        check(lib().gab_fmi_sa_lookup(self._h, _p(sm), C.c_int64(len(sm)), C.c_int32(max_occ), C.byref(co), C.byref(off), C.byref(tot)))
        n = tot.value
        coords = np.ctypeslib.as_array(C.cast(co, C.POINTER(C.c_int64)), shape=(max(n, 1),))[:n].copy()
        coff = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_int64)), shape=(len(sm) + 1,)).copy()
        lib().gab_fmi_free_coords(co); lib().gab_fmi_free_coords(off)
        return coords, coff

    def get_sa_entries_device(self, d_smems, n, max_occ, stream=0):
        """device pointer of gab_smem[n] (e.g. from seed_device) -> (device ptr coords, device ptr coord_off, total)"""
        co = C.c_void_p(); off = C.c_void_p(); tot = C.c_int64(0)
        check(lib().gab_fmi_sa_lookup_device(self._h, C.c_void_p(d_smems), C.c_int64(n), C.c_int32(max_occ), C.byref(co),
                                             C.byref(off), C.byref(tot), C.c_void_p(stream)))
        return co.value, off.value, tot.value

    def last_sa_stats(self):
        st = C.c_int64(0); ms = C.c_float(0)
        check(lib().gab_fmi_last_sa_stats(self._h, C.byref(st), C.byref(ms)))
        return {"lf_steps": st.value, "kernel_ms": ms.value}
