"""CPU: the wfa oracle (oracle/wfa.c) against golden CIGARs from the compiled reference."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_cigars


@pytest.mark.parametrize("name", ["wfa_bench", "wfa_adv"])
def test_oracle_matches_golden(name):
    batch = gabgen.read_pairs_text(f"{GOLDEN}/{name}.in.txt")
    want = read_cigars(f"{GOLDEN}/{name}.expected.txt")
    assert pyoracle.wfa_cigars(pyoracle.wfa(batch)) == want


@pytest.mark.parametrize("red", [(10, 10), (5, 3), (1, 0)])
def test_oracle_adaptive_matches_golden(red):
    """adaptive reduction (--minimum-wavefront-length / --maximum-difference-distance)"""
    batch = gabgen.read_pairs_text(f"{GOLDEN}/wfa_adv.in.txt")
    want = read_cigars(f"{GOLDEN}/wfa_adv.adaptive_{red[0]}_{red[1]}.expected.txt")
    got = pyoracle.wfa_cigars(pyoracle.wfa(batch, reduction=red))
    assert got == want
    assert got != read_cigars(f"{GOLDEN}/wfa_adv.expected.txt")       # the reduction really changes some alignments


def test_cigar_consistency():
    """size-independent property: the CIGAR consumes both strings and re-scores to the reported penalty"""
    b = gabgen.pairs(77, 3000, 1, 200)
    ops, off, ln, sc = pyoracle.wfa(b)
    for i in range(0, b.n, 7):
        o = bytes(ops[off[i]:off[i] + ln[i]])
        assert o.count(b"M") + o.count(b"X") + o.count(b"D") == b.pat_len[i]
        assert o.count(b"M") + o.count(b"X") + o.count(b"I") == b.txt_len[i]
        pen, prev = 0, b""
        for c in o:
            c = bytes([c])
            if c == b"X": pen += 4
            elif c in (b"I", b"D"): pen += 2 + (6 if c != prev else 0)
            prev = c
        assert pen == sc[i]


def test_edge_cases():
    b = gabgen.pairs_from_lists([b"A", b"ACGT", b"ACGT", b"AAAA", b"ACGTACGT", b"XXYY"], [b"A", b"ACGT", b"AGGT", b"AAAAAAAA", b"ACGT", b"YYXX"])
    got = pyoracle.wfa_cigars(pyoracle.wfa(b))
    assert got[0] == "1M" and got[1] == "4M" and got[2] == "1M1X2M"
