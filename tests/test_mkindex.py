"""CPU: the periodic-text index builder (tools/mkindex gab_mkindex_build_power) against the general builder, which is
itself byte-identical to `bwa-mem2 index` (tests/test_fmi_oracle.py).  The periodic builder is how the GPU tests get an
index of more than 2^32 rows -- intervals k, l, s that need the 40-bit arithmetic of a human-genome index -- in seconds."""
import numpy as np
import pytest

from tools import mkindex


def revcomp(a):
    return (3 - a[::-1]).astype(np.uint8)


@pytest.mark.parametrize("ulen,m,seed", [(7, 2, 1), (50, 2, 2), (333, 3, 3), (1000, 5, 4), (64, 16, 5), (1, 3, 6), (2, 4, 7)])
def test_power_index_equals_the_general_builder(ulen, m, seed):
    rng = np.random.default_rng(seed)
    U = rng.integers(0, 4, ulen).astype(np.uint8)
    if ulen <= 2:
        U = np.array([0, 1][:ulen], np.uint8)        # "A" -> W = "AT", "AC" -> W = "ACGT": primitive
    W = np.concatenate([U, revcomp(U)])
    ref = np.tile(W, m)
    assert np.array_equal(revcomp(ref), ref)             # its own reverse complement, so T = W^(2m)
    a = mkindex.FmIndex(ref)
    b = mkindex.FmIndex(U, power=m)
    assert a.ref_seq_len == b.ref_seq_len == 2 * len(ref) + 1
    assert np.array_equal(a.count, b.count)
    assert a.sentinel_index == b.sentinel_index
    assert np.array_equal(a.cp_occ, b.cp_occ)
    assert np.array_equal(a.sa_ms_byte, b.sa_ms_byte) and np.array_equal(a.sa_ls_word, b.sa_ls_word)


def test_power_index_rejects_a_non_primitive_word():
    U = np.array([0, 3, 0, 3], np.uint8)                 # U . revcomp(U) = ATATATAT = (AT)^4
    with pytest.raises(RuntimeError):
        mkindex.FmIndex(U, power=3)


def test_power_index_large_rows_consistent():
    """a few hundred million rows in well under a minute; the last checkpoint holds the totals"""
    rng = np.random.default_rng(11)
    U = rng.integers(0, 4, 200_000).astype(np.uint8)
    m = 300                                              # 2 * 300 * 400 000 = 240 M rows
    idx = mkindex.FmIndex(U, power=m)
    n = idx.ref_seq_len
    assert n == 2 * m * 400_000 + 1
    cp = idx.cp_occ.view(np.int64).reshape(-1, 8)
    last = cp[n >> 6]
    ones = np.array([bin(int(x) & (2**64 - 1)).count("1") for x in last[4:].view(np.uint64)])
    totals = last[:4] + ones
    assert np.array_equal(np.concatenate([[0], np.cumsum(totals)]), idx.count)
    assert (np.diff(cp[:, :4], axis=0) >= 0).all() and (np.diff(cp[:, :4].sum(axis=1)) <= 64).all()
