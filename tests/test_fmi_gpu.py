"""GPU parity: fmi HIP kernel (through the C ABI) vs the oracle and the golden output."""
import ctypes as C
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen, mkindex
from tests.util import GOLDEN, read_fasta_codes, read_fastq_reads

pytestmark = pytest.mark.gpu


def build(ref, tmp):
    idx = mkindex.FmIndex(ref)
    prefix = str(tmp / "ref")
    idx.write(prefix)
    return idx, prefix


def same(got, want):
    (g, goff), (w, woff) = got, want
    np.testing.assert_array_equal(goff, woff)
    assert len(g) == len(w)
    for f in ("rid", "m", "n", "k", "l", "s"):
        np.testing.assert_array_equal(g[f], w[f], err_msg=f)


def test_golden(tmp_path):
    from genarchbench_amd.fmi import FMI_search
    ref = read_fasta_codes(f"{GOLDEN}/fmi_small.ref.fa")
    idx, prefix = build(ref, tmp_path)
    reads = read_fastq_reads(f"{GOLDEN}/fmi_small.reads.fq")
    f = FMI_search(prefix)                      # load_index path: the .bwt.2bit.64 file
    sm, off = f.seed(reads, 19)
    assert pyoracle.fmi_text(sm, off) == open(f"{GOLDEN}/fmi_small.expected.txt").read()
    f.close()


@pytest.mark.parametrize("rseed,L,n,lo,hi,msl", [(81, 2_000_000, 60000, 151, 151, 19), (82, 300_000, 20000, 30, 250, 19),
                                                 (83, 50_000, 5000, 100, 151, 12), (84, 5_000, 3000, 20, 100, 25),
                                                 (85, 100_000, 1, 151, 151, 19)])
def test_vs_oracle(tmp_path, rseed, L, n, lo, hi, msl):
    from genarchbench_amd.fmi import FMI_search
    ref = gabgen.fmi_ref(rseed, L, 10)
    idx, prefix = build(ref, tmp_path)
    reads = gabgen.fmi_reads(rseed + 100, ref, n, lo, hi)
    f = FMI_search(arrays=(idx.ref_seq_len, idx.count, idx.cp_occ, idx.sentinel_index))   # in-memory path
    got = f.seed(reads, msl)
    oidx = pyoracle.fmi_load(prefix)
    w, woff, calls = pyoracle.fmi(oidx, reads, msl, want_calls=True)
    same(got, (w, woff))
    st = f.last_stats()
    assert st["ext_calls"] == calls and st["smems"] == len(w)
    f.close()


def test_seed_into_the_callers_array(tmp_path):
    """gab_fmi_seed_into (the reference's per-thread SMEM arrays, fmi.cpp:243-255,277-286) on a clone of the handle whose buffers
    gab_fmi_reserve sized beforehand: the records of each chunk of reads land in the caller's array, chunk-local read numbers;
    an array that is too small is refused with the number of records needed (GAB_ERANGE) and left to the caller"""
    from genarchbench_amd.fmi import FMI_search, SMEM_DTYPE
    from genarchbench_amd._lib import GabError
    ref = gabgen.fmi_ref(21, 300_000, 10)
    idx, prefix = build(ref, tmp_path)
    reads = gabgen.fmi_reads(22, ref, 9000, 80, 151)
    w, woff = pyoracle.fmi(pyoracle.fmi_load(prefix), reads, 19)
    owner = FMI_search(prefix)
    f = owner.clone()
    f.reserve(4000, reads.enc.shape[1])
    out = np.zeros(len(w) + 16, SMEM_DTYPE)
    used = 0
    for lo in range(0, 9000, 4000):                         # 4000 + 4000 + 1000 reads
        hi = min(9000, lo + 4000)
        n = f.seed_into(reads.enc[lo:hi], reads.len[lo:hi], out[used:], 19)
        assert n == woff[hi] - woff[lo]
        out["rid"][used:used + n] += np.uint32(lo)          # the driver's fix-up (fmi.cpp:340-343)
        used += n
    assert used == len(w)
    for fld in ("rid", "m", "n", "k", "l", "s"):
        np.testing.assert_array_equal(out[fld][:used], w[fld], err_msg=fld)
    small = np.zeros(10, SMEM_DTYPE)
    with pytest.raises(GabError) as e:
        f.seed_into(reads.enc[:4000], reads.len[:4000], small, 19)
    assert e.value.code == -34 and not small["s"].any()
    f.close(); owner.close()


def test_repetitive_reads_overflow_slots(tmp_path):
    """a highly repetitive reference makes some reads produce more SMEMs than the first-pass slot holds"""
    from genarchbench_amd.fmi import FMI_search
    rng = np.random.default_rng(4)
    unit = rng.integers(0, 4, 37).astype(np.uint8)
    ref = np.concatenate([np.tile(unit, 300), rng.integers(0, 4, 20000).astype(np.uint8), np.tile(unit[::-1], 200)])
    idx, prefix = build(ref, tmp_path)
    reads = gabgen.fmi_reads(9, ref, 3000, 300, 900)
    f = FMI_search(prefix)
    got = f.seed(reads, 10)
    w, woff = pyoracle.fmi(pyoracle.fmi_load(prefix), reads, 10)
    same(got, (w, woff))
    assert np.diff(woff).max() > 48
    f.close()


@pytest.mark.parametrize("lds_entries", [None, "4", "wide"])
def test_repetitive_short_reads_spill_lists(tmp_path, monkeypatch, lds_entries):
    """reads that fit the LDS path (<= 256 bases) on a repetitive reference: interval lists longer than the LDS ring
    (forced with a 4-entry ring in the second variant) spill to the global scratch and are fetched one step ahead;
    some reads also overflow their first-pass output slot.  Small indexes use the 13-byte list entries; the third variant
    forces the 16-byte format of indexes with 2^32 rows or more"""
    from genarchbench_amd.fmi import FMI_search
    rng = np.random.default_rng(6)
    unit = rng.integers(0, 4, 29).astype(np.uint8)
    ref = np.concatenate([np.tile(unit, 400), rng.integers(0, 4, 30000).astype(np.uint8), np.tile(3 - unit[::-1], 150)])
    idx, prefix = build(ref, tmp_path)
    reads = gabgen.fmi_reads(10, ref, 6000, 120, 250)
    w, woff, calls = pyoracle.fmi(pyoracle.fmi_load(prefix), reads, 10, want_calls=True)
    if lds_entries is not None:
        # the ring size is read once per process: exercise it through the C driver-style environment in a child
        import subprocess, sys, json, os
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from tools import gabgen; from genarchbench_amd.fmi import FMI_search;"
                "reads = gabgen.ReadBatch(np.load(%r), np.load(%r)); f = FMI_search(%r); sm, off = f.seed(reads, 10);"
                "np.save(%r, sm); np.save(%r, off); print(f.last_stats()['ext_calls'])")
        e, l = str(tmp_path / "enc.npy"), str(tmp_path / "len.npy")
        np.save(e, reads.enc); np.save(l, reads.len)
        so, oo = str(tmp_path / "sm.npy"), str(tmp_path / "off.npy")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, "-c", code % (root, e, l, prefix, so, oo)], capture_output=True, text=True,
                           env=dict(os.environ, GAB_FMI_WIDE_LISTS="1") if lds_entries == "wide" else
                           dict(os.environ, GAB_FMI_LDS_ENTRIES=lds_entries))
        assert r.returncode == 0, r.stderr[-2000:]
        got = (np.load(so), np.load(oo)); ext = int(r.stdout.split()[-1])
    else:
        f = FMI_search(prefix)
        got = f.seed(reads, 10); ext = f.last_stats()["ext_calls"]
        f.close()
    same(got, (w, woff))
    assert ext == calls


@pytest.mark.parametrize("wide, cap", [("0", None), ("4", None), ("4", "8"), ("2", "3")])
def test_wide_phases_handed_over(tmp_path, wide, cap):
    """backward phases whose lists stay wide leave the seeding kernel for fmi_wide_kernel (16 lanes per phase): on a repetitive
    reference with the threshold lowered to 4 / 2 survivors nearly every phase goes that way, pass-1 phases with their re-seeding
    candidates included; with room for 8 / 3 items and candidates the queues run full -- items stay with their lanes, a
    candidate without a place makes the library run the batch again without the hand-over; GAB_FMI_WIDE=0 is the kernel alone.
    Same SMEMs and the same number of extensions in every variant."""
    import subprocess, sys, os
    rng = np.random.default_rng(31)
    unit = rng.integers(0, 4, 41).astype(np.uint8)
    ref = np.concatenate([np.tile(unit, 120), rng.integers(0, 4, 40000).astype(np.uint8), np.tile(unit, 60), rng.integers(0, 4, 20000).astype(np.uint8)])
    idx, prefix = build(ref, tmp_path)
    reads = gabgen.fmi_reads(32, ref, 5000, 100, 151)
    w, woff, calls = pyoracle.fmi(pyoracle.fmi_load(prefix), reads, 19, want_calls=True)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from tools import gabgen; from genarchbench_amd.fmi import FMI_search;"
            "reads = gabgen.ReadBatch(np.load(%r), np.load(%r)); f = FMI_search(%r); sm, off = f.seed(reads, 19);"
            "np.save(%r, sm); np.save(%r, off); print(f.last_stats()['ext_calls'])")
    e, l = str(tmp_path / "enc.npy"), str(tmp_path / "len.npy")
    np.save(e, reads.enc); np.save(l, reads.len)
    so, oo = str(tmp_path / "sm.npy"), str(tmp_path / "off.npy")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAB_FMI_WIDE=wide)
    if cap:
        env["GAB_FMI_WIDE_CAP"] = cap
    r = subprocess.run([sys.executable, "-c", code % (root, e, l, prefix, so, oo)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    same((np.load(so), np.load(oo)), (w, woff))
    assert int(r.stdout.split()[-1]) == calls


def test_all_n_and_short_reads(tmp_path):
    from genarchbench_amd.fmi import FMI_search
    ref = gabgen.fmi_ref(5, 20000, 5)
    idx, prefix = build(ref, tmp_path)
    enc = np.full((6, 40), 4, np.uint8)
    enc[1, :40] = ref[100:140]; enc[2, :5] = ref[7:12]; enc[3, :40] = ref[300:340]; enc[3, 20] = 4
    enc[4, :1] = 2; enc[5, :40] = 3 - ref[500:540][::-1]
    reads = gabgen.ReadBatch(enc, np.array([40, 40, 5, 40, 1, 40], np.int32))
    f = FMI_search(prefix)
    same(f.seed(reads, 19), pyoracle.fmi(pyoracle.fmi_load(prefix), reads, 19))
    f.close()


def test_device_resident(tmp_path):
    import ctypes as C
    import torch
    from genarchbench_amd.fmi import FMI_search, SMEM_DTYPE
    ref = gabgen.fmi_ref(91, 400000, 5)
    idx, prefix = build(ref, tmp_path)
    reads = gabgen.fmi_reads(92, ref, 30000, 151, 151)
    f = FMI_search(prefix)
    dev = torch.device("cuda:0")
    enc = torch.from_numpy(reads.enc).to(dev); ln = torch.from_numpy(reads.len).to(dev)
    d_out, d_off, n = f.seed_device(enc, ln, 19, stream=torch.cuda.current_stream().cuda_stream)
    w, woff = pyoracle.fmi(pyoracle.fmi_load(prefix), reads, 19)
    assert n == len(w)
    host = np.zeros(n * 40, np.uint8)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(d_out), C.c_size_t(n * 40), C.c_int(2)) == 0
    off = np.zeros(reads.n + 1, np.int64)
    assert hip.hipMemcpy(off.ctypes.data_as(C.c_void_p), C.c_void_p(d_off), C.c_size_t(8 * (reads.n + 1)), C.c_int(2)) == 0
    same((host.view(SMEM_DTYPE), off), (w, woff))
    f.close()


# ---- suffix-array look-up -------------------------------------------------------------------
@pytest.mark.parametrize("rseed,L,n,max_occ", [(91, 400_000, 20000, 500), (92, 400_000, 20000, 3), (93, 30_000, 4000, 1),
                                                (94, 3_000, 2000, 50)])
def test_sa_lookup_vs_oracle(tmp_path, rseed, L, n, max_occ):
    from genarchbench_amd.fmi import FMI_search
    ref = gabgen.fmi_ref(rseed, L, 10)
    idx, prefix = build(ref, tmp_path)
    reads = gabgen.fmi_reads(rseed + 100, ref, n, 40, 151)
    f = FMI_search(prefix=prefix)                       # gab_fmi_load reads the sampled suffix array from the file
    sm, _ = f.seed(reads, 19)
    got, goff = f.get_sa_entries(sm, max_occ)
    oidx = pyoracle.fmi_load(prefix)
    want, woff, steps = pyoracle.fmi_sa_lookup(oidx, sm, max_occ)
    np.testing.assert_array_equal(goff, woff)
    np.testing.assert_array_equal(got, want)
    assert f.last_sa_stats()["lf_steps"] == steps
    f.close()


def test_sa_lookup_in_memory_index_and_errors():
    from genarchbench_amd._lib import GabError
    from genarchbench_amd.fmi import FMI_search, SMEM_DTYPE
    ref = gabgen.fmi_ref(95, 50_000, 5)
    idx = mkindex.FmIndex(ref)
    f = FMI_search(arrays=(idx.ref_seq_len, idx.count, idx.cp_occ, idx.sentinel_index))
    reads = gabgen.fmi_reads(96, ref, 500, 80, 151)
    sm, _ = f.seed(reads, 19)
    with pytest.raises(GabError):
        f.get_sa_entries(sm, 10)                        # no suffix array attached yet
    f.set_sa(idx.sa_ms_byte, idx.sa_ls_word)
    got, goff = f.get_sa_entries(sm, 10)
    oidx = pyoracle.FmIndex()
    cnt = (C.c_int64 * 5)(*[int(x) for x in idx.count])
    pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(idx.ref_seq_len), cnt, idx.cp_occ.ctypes.data_as(C.c_void_p),
                                          C.c_int64(idx.sentinel_index))
    pyoracle.lib().oracle_fmi_set_sa(C.byref(oidx), idx.sa_ms_byte.ctypes.data_as(C.c_void_p), idx.sa_ls_word.ctypes.data_as(C.c_void_p))
    want, woff, _ = pyoracle.fmi_sa_lookup(oidx, sm, 10)
    np.testing.assert_array_equal(goff, woff)
    np.testing.assert_array_equal(got, want)
    # empty input and the sentinel row (k = sentinel index, s = 1 -> coordinate 0 + walk length)
    e, eoff = f.get_sa_entries(np.zeros(0, SMEM_DTYPE), 10)
    assert len(e) == 0 and list(eoff) == [0]
    one = np.zeros(3, SMEM_DTYPE); one["k"] = [idx.sentinel_index, 0, 8]; one["s"] = [1, 1, 3]
    g2, o2 = f.get_sa_entries(one, 10)
    w2, wo2, _ = pyoracle.fmi_sa_lookup(oidx, one, 10)
    np.testing.assert_array_equal(g2, w2); np.testing.assert_array_equal(o2, wo2)
    f.close()


# ---- an index of more than 2^32 rows (a human genome has ~6.2 G): k, l, s need the 40-bit forms everywhere ------------
def _oracle_index(idx, with_sa=False):
    oidx = pyoracle.FmIndex()
    cnt = (C.c_int64 * 5)(*[int(x) for x in idx.count])
    pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(idx.ref_seq_len), cnt, idx.cp_occ.ctypes.data_as(C.c_void_p),
                                          C.c_int64(idx.sentinel_index))
    if with_sa:
        pyoracle.lib().oracle_fmi_set_sa(C.byref(oidx), idx.sa_ms_byte.ctypes.data_as(C.c_void_p), idx.sa_ls_word.ctypes.data_as(C.c_void_p))
    return oidx


def test_index_with_more_than_2_32_rows():
    """The reference (U . revcomp(U))^m is indexed without sorting the whole text (tools/mkindex, checked against the
    general builder in tests/test_mkindex.py): 4.4 G rows here, so interval bounds above 2^32 really flow through the
    seeding kernel -- the 16-byte LDS list entries, pack / unpack of k, l, s, the short-pattern table, the output records --
    and through the suffix-array look-up (coordinates above 2^32: the sampled SA's high bytes)."""
    from genarchbench_amd.fmi import FMI_search
    U = gabgen.fmi_ref(77, 1_000_000, 5)
    m = 1101                                                     # k = 2202 copies of the 2 Mbp word
    idx = mkindex.FmIndex(U, power=m)
    assert idx.ref_seq_len > 2 ** 32
    W = np.concatenate([U, (3 - U[::-1]).astype(np.uint8)])
    reads = gabgen.fmi_reads(78, np.concatenate([W, W]), 40000, 60, 151)      # substitutions, N's, both strands
    f = FMI_search(arrays=(idx.ref_seq_len, idx.count, idx.cp_occ, idx.sentinel_index))
    got = f.seed(reads, 19)
    oidx = _oracle_index(idx, with_sa=True)
    w, woff, calls = pyoracle.fmi(oidx, reads, 19, want_calls=True)
    same(got, (w, woff))
    assert f.last_stats()["ext_calls"] == calls
    assert (w["k"] > 2 ** 32).sum() > 1000 and (w["l"] > 2 ** 32).sum() > 1000 and w["s"].min() >= 2 * m
    # the step after seeding on the same index: a few SMEMs only (LF walks are long in a periodic text)
    f.set_sa(idx.sa_ms_byte, idx.sa_ls_word)
    sub = w[(w["k"] > 2 ** 32)][:60]
    co, coff = f.get_sa_entries(sub, 3)
    wco, wcoff, _ = pyoracle.fmi_sa_lookup(oidx, sub, 3)
    np.testing.assert_array_equal(coff, wcoff)
    np.testing.assert_array_equal(co, wco)
    assert (wco > 2 ** 32).any()
    f.close()


def test_scratch_budget_is_honoured(tmp_path, monkeypatch):
    """$GAB_FMI_SCRATCH_MB: long reads (no LDS lists, every list entry in the per-lane spill area) run on fewer resident
    waves instead of a spill area of stride x 32 B x every lane of the chip, and a repetitive read that overflows its output
    slot shrinks the batches instead of multiplying the slot buffer; results unchanged"""
    from genarchbench_amd.fmi import FMI_search
    rng = np.random.default_rng(8)
    unit = rng.integers(0, 4, 37).astype(np.uint8)
    ref = np.concatenate([np.tile(unit, 300), rng.integers(0, 4, 20000).astype(np.uint8), np.tile(unit[::-1], 200)])
    idx, prefix = build(ref, tmp_path)
    long_reads = gabgen.fmi_reads(12, ref, 30000, 300, 1500)                 # stride 1500: the global-list path
    short_reads = gabgen.fmi_reads(13, ref, 40000, 120, 250)
    want_long = pyoracle.fmi(pyoracle.fmi_load(prefix), long_reads, 10)
    want_short = pyoracle.fmi(pyoracle.fmi_load(prefix), short_reads, 10)
    monkeypatch.setenv("GAB_FMI_SCRATCH_MB", "64")
    f = FMI_search(prefix)
    same(f.seed(long_reads, 10), want_long)
    same(f.seed(short_reads, 10), want_short)
    # some read overflowed the first-pass slot, and the enlarged slots of the whole batch would not fit the budget
    assert np.diff(want_long[1]).max() > 48 and (np.diff(want_long[1]).max() + 8) * 32 * 30000 > 64 << 20
    f.close()
