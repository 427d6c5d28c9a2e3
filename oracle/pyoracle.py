"""ctypes front-end for the CPU ORACLE (oracle/*.c) -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product path (genarchbench_amd/, benchmarks/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build(force=False, with_ref=True):
    """compile liboracle.so and, when /root/reference is present, oracle/_ref/*"""
    targets = ["all"] + (["ref"] if with_ref else [])
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale or (with_ref and os.path.isdir("/root/reference")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + targets)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build(with_ref=False))
    return _lib


def ref_path(name):
    """path of a compiled-reference binary under oracle/_ref, or None if it was not built"""
    p = os.path.join(_HERE, "_ref", name)
    return p if os.path.exists(p) else None


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ bsw
class BswParams(C.Structure):
    _fields_ = [("o_del", C.c_int32), ("e_del", C.c_int32), ("o_ins", C.c_int32),
                ("e_ins", C.c_int32), ("zdrop", C.c_int32), ("end_bonus", C.c_int32),
                ("w", C.c_int32), ("mat", C.c_int8 * 25)]


BSW_RESULT_FIELDS = ("score", "qle", "tle", "gtle", "gscore", "max_off")


def bsw_params(match=1, mismatch=4, gapo=6, gape=1, ambig=-1, zdrop=100, end_bonus=5, w=100):
    """defaults of the reference driver: bsw/src/main_banded.cpp:70-74,266-276"""
    p = BswParams(gapo, gape, gapo, gape, zdrop, end_bonus, w)
    lib().oracle_bsw_fill_scmat(C.c_int(match), C.c_int(mismatch), C.c_int(ambig), p.mat)
    return p


def bsw(batch, params=None, threads=0, want_cells=False):
    """returns int32 array [n, 6] = (score, qle, tle, gtle, gscore, max_off)"""
    params = params or bsw_params()
    out = np.zeros((batch.n, 6), np.int32)
    cells = C.c_int64(0)
    lib().oracle_bsw_batch(C.byref(params), _p(batch.ref), _p(batch.ref_off), _p(batch.qry),
                           _p(batch.qry_off), _p(batch.len1), _p(batch.len2), _p(batch.h0),
                           C.c_int64(batch.n), C.c_int(threads), _p(out), C.byref(cells))
    return (out, cells.value) if want_cells else out


# ------------------------------------------------------------------ chain / fast-chain
def chain(batch, mode=0, threads=0, want_evals=False):
    """mode 0 = chain (max_skip heuristics), 1 = fast-chain (AVX2/AVX-512 arithmetic).
    returns (score, parent) int32 arrays over all anchors of all calls"""
    score = np.zeros(batch.nanchors, np.int32); parent = np.zeros(batch.nanchors, np.int32)
    ev = C.c_int64(0)
    lib().oracle_chain_batch(C.c_int(mode), _p(batch.hdr), _p(batch.call_off), C.c_int64(batch.ncalls),
                             _p(batch.x), _p(batch.y), C.c_int(threads), _p(score), _p(parent), C.byref(ev))
    return (score, parent, ev.value) if want_evals else (score, parent)


# ------------------------------------------------------------------ bpm
def bpm(batch, threads=0, want_steps=False):
    """batch: PairBatch with the longer-is-pattern swap already applied -> int32 scores (<= 0)"""
    score = np.zeros(batch.n, np.int32)
    st = C.c_int64(0)
    lib().oracle_bpm_batch(_p(batch.pat), _p(batch.pat_off), _p(batch.pat_len), _p(batch.txt), _p(batch.txt_off),
                           _p(batch.txt_len), C.c_int64(batch.n), C.c_int(threads), _p(score), C.byref(st))
    return (score, st.value) if want_steps else score


# ------------------------------------------------------------------ bitpal
def bitpal(batch, algorithm, threads=0):
    """algorithm 0 = bitpal-edit, 1 = bitpal-scored -> printed scores (int32)"""
    out = np.zeros(batch.n, np.int32)
    lib().oracle_bitpal_batch(C.c_int(algorithm), _p(batch.pat), _p(batch.pat_off), _p(batch.pat_len), _p(batch.txt),
                              _p(batch.txt_off), _p(batch.txt_len), C.c_int64(batch.n), C.c_int(threads), _p(out))
    return out


# ------------------------------------------------------------------ wfa
class WfaPenalties(C.Structure):
    _fields_ = [("mismatch", C.c_int32), ("gap_opening", C.c_int32), ("gap_extension", C.c_int32),
                ("min_wavefront_length", C.c_int32), ("max_distance_threshold", C.c_int32)]


def rle(ops):
    """edit_cigar_print (wfa/gap_affine/edit_cigar.c:184-200): run-length "%d%c" text"""
    if len(ops) == 0:
        return ""
    a = np.frombuffer(ops, np.uint8) if isinstance(ops, (bytes, bytearray)) else np.asarray(ops, np.uint8)
    cut = np.flatnonzero(np.diff(a)) + 1
    starts = np.concatenate([[0], cut]); ends = np.concatenate([cut, [len(a)]])
    return "".join("%d%c" % (e - s, a[s]) for s, e in zip(starts, ends))


def wfa(batch, pen=(4, 6, 2), threads=0, want_cells=False, reduction=None):
    """batch: PairBatch (no swap: '>' line is the pattern).  returns (ops slab, ops_off, ops_len, score).
    reduction = (min_wavefront_length, max_distance_threshold) selects the adaptive mode"""
    p = WfaPenalties(*pen, *(reduction if reduction is not None else (-1, -1)))
    cap = batch.pat_len.astype(np.int64) + batch.txt_len.astype(np.int64)
    ops_off = np.zeros(batch.n, np.int64)
    if batch.n > 1:
        np.cumsum(cap[:-1], out=ops_off[1:])
    ops = np.zeros(int(cap.sum()) + 16, np.uint8)
    ops_len = np.zeros(batch.n, np.int32); score = np.zeros(batch.n, np.int32)
    cells = C.c_int64(0)
    lib().oracle_wfa_batch(C.byref(p), _p(batch.pat), _p(batch.pat_off), _p(batch.pat_len), _p(batch.txt),
                           _p(batch.txt_off), _p(batch.txt_len), C.c_int64(batch.n), C.c_int(threads),
                           _p(ops), _p(ops_off), _p(ops_len), _p(score), C.byref(cells))
    r = (ops, ops_off, ops_len, score)
    return r + (cells.value,) if want_cells else r


def wfa_cigars(res, idx=None):
    ops, off, ln = res[0], res[1], res[2]
    idx = range(len(ln)) if idx is None else idx
    return [rle(ops[off[i]:off[i] + ln[i]]) for i in idx]


# ------------------------------------------------------------------ fmi
class FmIndex(C.Structure):
    _fields_ = [("ref_seq_len", C.c_int64), ("count", C.c_int64 * 5), ("cp_occ_size", C.c_int64),
                ("sentinel_index", C.c_int64), ("cp_occ", C.c_void_p), ("sa_ms_byte", C.c_void_p), ("sa_ls_word", C.c_void_p)]


SMEM_DTYPE = np.dtype([("rid", np.uint32), ("m", np.uint32), ("n", np.uint32), ("pad", np.uint32),
                       ("k", np.int64), ("l", np.int64), ("s", np.int64)])
assert SMEM_DTYPE.itemsize == 40


def fmi_load(prefix):
    idx = FmIndex()
    rc = lib().oracle_fmi_load(prefix.encode(), C.byref(idx))
    if rc:
        raise IOError(f"cannot load {prefix}.bwt.2bit.64 ({rc})")
    return idx


def fmi(idx, reads, min_seed_len=19, threads=0, want_calls=False):
    """returns (smems structured array sorted by rid/m/n desc, read_off int64[n+1])"""
    L = lib()
    L.oracle_fmi_batch.restype = C.c_int64
    out = C.c_void_p(); off = np.zeros(reads.n + 1, np.int64); calls = C.c_int64(0)
    n = L.oracle_fmi_batch(C.byref(idx), _p(reads.enc), C.c_int32(reads.stride), _p(reads.len), C.c_int64(reads.n),
                           C.c_int(min_seed_len), C.c_int(threads), C.byref(out), _p(off), C.byref(calls))
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(n, 1) * 40,))[:n * 40].copy().view(SMEM_DTYPE)
    L.oracle_fmi_release(out)
    return (arr, off, calls.value) if want_calls else (arr, off)


def fmi_sa_lookup(idx, smems, max_occ):
    """suffix-array coordinates of SMEM intervals -> (coords int64[total], coord_off int64[n+1], lf_steps)"""
    L = lib()
    L.oracle_fmi_sa_count.restype = C.c_int64; L.oracle_fmi_sa_lookup.restype = C.c_int64
    sm = np.ascontiguousarray(smems)
    off = np.zeros(len(sm) + 1, np.int64)
    total = L.oracle_fmi_sa_count(_p(sm), C.c_int64(len(sm)), C.c_int32(max_occ), _p(off))
    coords = np.zeros(max(total, 1), np.int64); steps = C.c_int64(0)
    rc = L.oracle_fmi_sa_lookup(C.byref(idx), _p(sm), C.c_int64(len(sm)), C.c_int32(max_occ), _p(off), _p(coords), C.byref(steps))
    if rc < 0:
        raise ValueError("the index has no suffix-array samples")
    return coords[:total], off, steps.value


def fmi_text(smems, read_off):
    """the data part of the reference's stdout (fmi.cpp:430-460): "rid:" then "[m,n+1]" per SMEM, for every
    read id up to the last read that has a seed"""
    lines = []
    last = 0
    for r in range(len(read_off) - 1):
        if read_off[r + 1] > read_off[r]:
            last = r
    if len(smems) == 0:
        return ""
    for r in range(last + 1):
        lines.append(f"{r}:")
        for j in range(read_off[r], read_off[r + 1]):
            lines.append(f"[{smems['m'][j]},{smems['n'][j] + 1}]")
    return "\n".join(lines) + "\n"
