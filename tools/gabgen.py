"""ctypes front-end for the seeded input generators in tools/gen (test/bench infrastructure)."""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_GEN_DIR = os.path.join(_HERE, "gen")
_lib = None


def build(force=False):
    so = os.path.join(_GEN_DIR, "libgabgen.so")
    exe = os.path.join(_GEN_DIR, "gabgen")
    srcs = [os.path.join(_GEN_DIR, f) for f in os.listdir(_GEN_DIR) if f.endswith((".c", ".h"))]
    stale = not (os.path.exists(so) and os.path.exists(exe)) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs)
    if force or stale:
        import fcntl
        with open(os.path.join(_GEN_DIR, ".build.lock"), "w") as lk:       # several ranks / test workers may get here at once
            fcntl.flock(lk, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", _GEN_DIR, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _offsets(lens, align=1):
    """exclusive prefix sum of lens (each rounded up to `align`) -> int64 offsets, total"""
    l = lens.astype(np.int64)
    if align > 1:
        l = (l + align - 1) // align * align
    off = np.zeros(len(l), dtype=np.int64)
    if len(l) > 1:
        np.cumsum(l[:-1], out=off[1:])
    total = int(off[-1] + l[-1]) if len(l) else 0
    return off, total


@dataclass
class BswBatch:
    """Packed bsw input: codes 0..4, one byte per base, variable-length, back to back."""
    ref: np.ndarray
    ref_off: np.ndarray
    qry: np.ndarray
    qry_off: np.ndarray
    len1: np.ndarray  # reference (target) lengths
    len2: np.ndarray  # query lengths
    h0: np.ndarray

    @property
    def n(self):
        return len(self.len1)

    def pair(self, i):
        r = self.ref[self.ref_off[i]:self.ref_off[i] + self.len1[i]]
        q = self.qry[self.qry_off[i]:self.qry_off[i] + self.len2[i]]
        return r, q, int(self.h0[i])

    def write_text(self, path):
        """reference input format: bsw/src/main_banded.cpp:152-206"""
        with open(path, "wb") as f:
            for i in range(self.n):
                r, q, h = self.pair(i)
                f.write(b"%d\n" % h)
                f.write((r + 48).astype(np.uint8).tobytes() + b"\n")
                f.write((q + 48).astype(np.uint8).tobytes() + b"\n")


def bsw(seed, n, mode=0, first=0):
    L = lib()
    len1 = np.empty(n, np.int32); len2 = np.empty(n, np.int32); h0 = np.empty(n, np.int32)
    L.gab_gen_bsw_lens(C.c_uint64(seed), C.c_int(mode), C.c_int64(first), C.c_int64(n),
                       _p(len1), _p(len2), _p(h0))
    ref_off, rt = _offsets(len1)
    qry_off, qt = _offsets(len2)
    # 16 bytes of slack so device-side vector loads past the last base stay in bounds
    ref = np.zeros(rt + 16, np.uint8); qry = np.zeros(qt + 16, np.uint8)
    L.gab_gen_bsw_fill(C.c_uint64(seed), C.c_int(mode), C.c_int64(first), C.c_int64(n),
                       _p(ref), _p(ref_off), _p(qry), _p(qry_off))
    return BswBatch(ref, ref_off, qry, qry_off, len1, len2, h0)


def bsw_from_arrays(refs, qrys, h0s):
    """build a batch from python lists of uint8 code arrays (hand-made edge cases)"""
    len1 = np.array([len(r) for r in refs], np.int32)
    len2 = np.array([len(q) for q in qrys], np.int32)
    ref_off, rt = _offsets(len1); qry_off, qt = _offsets(len2)
    ref = np.zeros(rt + 16, np.uint8); qry = np.zeros(qt + 16, np.uint8)
    for i, (r, q) in enumerate(zip(refs, qrys)):
        ref[ref_off[i]:ref_off[i] + len(r)] = r
        qry[qry_off[i]:qry_off[i] + len(q)] = q
    return BswBatch(ref, ref_off, qry, qry_off, len1, len2, np.array(h0s, np.int32))


def write_text(bench, path, seed, n, mode=0, *extra):
    build()
    subprocess.check_call([os.path.join(_GEN_DIR, "gabgen"), bench, path, str(seed), str(n), str(mode)]
                          + [str(e) for e in extra])


# ------------------------------------------------------------------ chain / fast-chain
CHAIN_HDR = np.dtype([("n", np.int64), ("avg_qspan", np.float32), ("max_dist_x", np.int32),
                      ("max_dist_y", np.int32), ("bw", np.int32), ("n_segs", np.int32)], align=True)
assert CHAIN_HDR.itemsize == 32


@dataclass
class ChainBatch:
    """all calls of one input file, anchors back to back: call c owns [call_off[c], call_off[c]+hdr[c].n)"""
    hdr: np.ndarray       # CHAIN_HDR records
    call_off: np.ndarray  # int64
    x: np.ndarray         # uint64
    y: np.ndarray         # uint64

    @property
    def ncalls(self):
        return len(self.hdr)

    @property
    def nanchors(self):
        return len(self.x)

    def write_text(self, path):
        """reference input format: chain/src/host_data_io.cpp:13-51"""
        with open(path, "w") as f:
            for c in range(self.ncalls):
                h = self.hdr[c]; o = int(self.call_off[c])
                f.write("%d\t%f\t%d\t%d\t%d\t%d\n" % (h["n"], h["avg_qspan"], h["max_dist_x"], h["max_dist_y"],
                                                     h["bw"], h["n_segs"]))
                for i in range(int(h["n"])):
                    f.write("%d\t%d\n" % (self.x[o + i], self.y[o + i]))
                f.write("EOR\n")


def chain(seed, ncalls, mode=0, nmin=50, nmax=60000, first=0):
    L = lib()
    hdr = np.zeros(ncalls, CHAIN_HDR)
    L.gab_gen_chain_hdrs(C.c_uint64(seed), C.c_int(mode), C.c_int64(nmin), C.c_int64(nmax), C.c_int64(first),
                         C.c_int64(ncalls), _p(hdr))
    call_off, total = _offsets(hdr["n"])
    x = np.zeros(total, np.uint64); y = np.zeros(total, np.uint64)
    L.gab_gen_chain_fill(C.c_uint64(seed), C.c_int(mode), C.c_int64(nmin), C.c_int64(nmax), C.c_int64(first),
                         C.c_int64(ncalls), _p(hdr), _p(call_off), _p(x), _p(y))
    return ChainBatch(hdr, call_off, x, y)


def chain_sizes(seed, ncalls, mode=0, nmin=50, nmax=60000):
    """anchor count of calls 0 .. ncalls-1 (headers only: cheap)"""
    hdr = np.zeros(ncalls, CHAIN_HDR)
    lib().gab_gen_chain_hdrs(C.c_uint64(seed), C.c_int(mode), C.c_int64(nmin), C.c_int64(nmax), C.c_int64(0), C.c_int64(ncalls), _p(hdr))
    return hdr["n"].copy()


def chain_ids(seed, ids, mode=0, nmin=50, nmax=60000):
    """the calls with the given ids (any subset, any order) of the same seeded input chain(seed, N, ...) generates"""
    L = lib()
    ids = np.ascontiguousarray(ids, np.int64)
    hdr = np.zeros(len(ids), CHAIN_HDR)
    L.gab_gen_chain_ids(C.c_uint64(seed), C.c_int(mode), C.c_int64(nmin), C.c_int64(nmax), _p(ids), C.c_int64(len(ids)), _p(hdr))
    call_off, total = _offsets(hdr["n"])
    x = np.zeros(total, np.uint64); y = np.zeros(total, np.uint64)
    L.gab_gen_chain_fill_ids(C.c_uint64(seed), C.c_int(mode), _p(ids), C.c_int64(len(ids)), _p(hdr), _p(call_off), _p(x), _p(y))
    return ChainBatch(hdr, call_off, x, y)


def chain_from_calls(calls):
    """calls: list of (avg_qspan, max_dist_x, max_dist_y, bw, n_segs, x_array, y_array)"""
    hdr = np.zeros(len(calls), CHAIN_HDR)
    for c, (aq, mdx, mdy, bw, ns, x, y) in enumerate(calls):
        hdr[c] = (len(x), aq, mdx, mdy, bw, ns)
    call_off, total = _offsets(hdr["n"])
    X = np.zeros(total, np.uint64); Y = np.zeros(total, np.uint64)
    for c, call in enumerate(calls):
        o = int(call_off[c]); n = len(call[5])
        X[o:o + n] = call[5]; Y[o:o + n] = call[6]
    return ChainBatch(hdr, call_off, X, Y)


def read_chain_text(path):
    """parse the reference's chain input format into a ChainBatch (tests only; the C driver has its own parser)"""
    toks = open(path).read().split()
    calls, p = [], 0
    while p + 6 <= len(toks):
        n = int(toks[p]); aq = np.float32(toks[p + 1])
        mdx, mdy, bw, ns = (int(t) for t in toks[p + 2:p + 6])
        p += 6
        xy = np.array(toks[p:p + 2 * n], dtype=np.uint64).reshape(n, 2)
        p += 2 * n
        assert toks[p] == "EOR"
        p += 1
        calls.append((aq, mdx, mdy, bw, ns, xy[:, 0].copy(), xy[:, 1].copy()))
    return chain_from_calls(calls)


# ------------------------------------------------------------------ bpm / wfa
@dataclass
class PairBatch:
    """'>' pattern / '<' text pairs, ASCII, back to back"""
    pat: np.ndarray
    pat_off: np.ndarray
    pat_len: np.ndarray
    txt: np.ndarray
    txt_off: np.ndarray
    txt_len: np.ndarray

    @property
    def n(self):
        return len(self.pat_len)

    def pair(self, i):
        p = self.pat[self.pat_off[i]:self.pat_off[i] + self.pat_len[i]].tobytes()
        t = self.txt[self.txt_off[i]:self.txt_off[i] + self.txt_len[i]].tobytes()
        return p, t

    def swapped_longer_first(self):
        """the bpm driver's swap (bpm/tools/align_benchmark.c:177-181): the longer LINE becomes the pattern"""
        swap = self.pat_len < self.txt_len
        if not swap.any():
            return self
        pats, txts = [], []
        for i in range(self.n):
            p, t = self.pair(i)
            if swap[i]:
                p, t = t, p
            pats.append(p); txts.append(t)
        return pairs_from_lists(pats, txts)

    def swapped_combined(self):
        """vectorised form of the swap for big batches: ONE slab holding [patterns | texts] and
        offset/length arrays pointing into it, roles exchanged where the text line is longer"""
        slab = np.concatenate([self.pat, self.txt])
        base = np.int64(len(self.pat))
        swap = self.pat_len < self.txt_len
        po = np.where(swap, self.txt_off + base, self.pat_off).astype(np.int64)
        to = np.where(swap, self.pat_off, self.txt_off + base).astype(np.int64)
        pl = np.where(swap, self.txt_len, self.pat_len).astype(np.int32)
        tl = np.where(swap, self.pat_len, self.txt_len).astype(np.int32)
        return PairBatch(slab, po, pl, slab, to, tl)

    def interleaved(self, gap=2):
        """the same pairs in ONE slab laid out like the drivers' input file: pattern i, text i, pattern i + 1, ... with
        `gap` bytes between sequences (the file's "\\n<" / "\\n>"); the window of a chunk of pairs is then contiguous"""
        L = lib()
        n = self.n
        po = np.empty(n, np.int64); to = np.empty(n, np.int64)
        args = [_p(self.pat), _p(self.pat_off), _p(self.pat_len), _p(self.txt), _p(self.txt_off), _p(self.txt_len), C.c_int64(n), C.c_int(gap)]
        L.gab_gen_interleave.restype = C.c_int64
        total = L.gab_gen_interleave(*args, None, _p(po), _p(to))
        slab = np.zeros(total + 16, np.uint8)
        L.gab_gen_interleave(*args, _p(slab), _p(po), _p(to))
        return PairBatch(slab, po, self.pat_len, slab, to, self.txt_len)

    def write_text(self, path):
        with open(path, "wb") as f:
            for i in range(self.n):
                p, t = self.pair(i)
                f.write(b">" + p + b"\n<" + t + b"\n")


def pairs(seed, n, mode=0, plen=151, first=0):
    L = lib()
    pl = np.empty(n, np.int32); tl = np.empty(n, np.int32)
    L.gab_gen_pairs_lens(C.c_uint64(seed), C.c_int(mode), C.c_int(plen), C.c_int64(first), C.c_int64(n), _p(pl), _p(tl))
    po, pt = _offsets(pl); to, tt = _offsets(tl)
    pat = np.zeros(pt + 16, np.uint8); txt = np.zeros(tt + 16, np.uint8)
    L.gab_gen_pairs_fill(C.c_uint64(seed), C.c_int(mode), C.c_int(plen), C.c_int64(first), C.c_int64(n),
                         _p(pat), _p(po), _p(txt), _p(to))
    return PairBatch(pat, po, pl, txt, to, tl)


def pairs_from_lists(pats, txts):
    pl = np.array([len(p) for p in pats], np.int32); tl = np.array([len(t) for t in txts], np.int32)
    po, pt = _offsets(pl); to, tt = _offsets(tl)
    pat = np.zeros(pt + 16, np.uint8); txt = np.zeros(tt + 16, np.uint8)
    for i, (p, t) in enumerate(zip(pats, txts)):
        pat[po[i]:po[i] + len(p)] = np.frombuffer(p, np.uint8)
        txt[to[i]:to[i] + len(t)] = np.frombuffer(t, np.uint8)
    return PairBatch(pat, po, pl, txt, to, tl)


def read_pairs_text(path):
    lines = open(path, "rb").read().split(b"\n")
    pats = [l[1:] for l in lines[0::2] if l]
    txts = [l[1:] for l in lines[1::2] if l]
    return pairs_from_lists(pats, txts)


# ------------------------------------------------------------------ fmi
def fmi_ref(seed, ref_len, rep_pct=5):
    ref = np.zeros(ref_len, np.uint8)
    lib().gab_gen_fmi_ref(C.c_uint64(seed), C.c_int64(ref_len), C.c_int(rep_pct), _p(ref))
    return ref


@dataclass
class ReadBatch:
    enc: np.ndarray     # [n, stride] uint8 codes 0..4 (what fmi.cpp:121-151 builds)
    len: np.ndarray     # int32

    @property
    def n(self):
        return len(self.len)

    @property
    def stride(self):
        return self.enc.shape[1]


def fmi_reads(seed, ref, n, rl_min=151, rl_max=151, first=0):
    enc = np.zeros((n, rl_max), np.uint8); ln = np.zeros(n, np.int32)
    lib().gab_gen_fmi_reads(C.c_uint64(seed), _p(ref), C.c_int64(len(ref)), C.c_int(rl_min), C.c_int(rl_max),
                            C.c_int64(first), C.c_int64(n), _p(enc), C.c_int32(rl_max), _p(ln))
    return ReadBatch(enc, ln)


def fmi_write_fasta(path, ref):
    assert lib().gab_gen_fmi_write_fasta(path.encode(), _p(ref), C.c_int64(len(ref))) == 0


def fmi_write_fastq(path, reads):
    assert lib().gab_gen_fmi_write_fastq(path.encode(), _p(reads.enc), C.c_int32(reads.stride), _p(reads.len),
                                         C.c_int64(reads.n)) == 0
