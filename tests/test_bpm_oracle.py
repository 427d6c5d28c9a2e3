"""CPU: the bpm oracle (oracle/bpm.c) against golden vectors from the compiled reference."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_scores


def lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


@pytest.mark.parametrize("name", ["bpm_bench", "bpm_adv"])
def test_oracle_matches_golden(name):
    batch = gabgen.read_pairs_text(f"{GOLDEN}/{name}.in.txt").swapped_longer_first()
    want = read_scores(f"{GOLDEN}/{name}.expected.txt")
    np.testing.assert_array_equal(pyoracle.bpm(batch), want)


def test_adv_fixture_contains_non_levenshtein_cases():
    """the N-aliasing / raw-compare quirks must be exercised: some printed scores differ from -Levenshtein"""
    batch = gabgen.read_pairs_text(f"{GOLDEN}/bpm_adv.in.txt").swapped_longer_first()
    want = read_scores(f"{GOLDEN}/bpm_adv.expected.txt")
    diff = clean_ok = 0
    for i in range(0, batch.n, 5):
        p, t = batch.pair(i)
        d = -lev(p, t)
        if set(p + t) <= set(b"ACGT"):
            assert want[i] == d          # clean pairs: exactly the edit distance
            clean_ok += 1
        elif want[i] != d:
            diff += 1
    assert diff > 5 and clean_ok > 5


def test_edge_cases():
    b = gabgen.pairs_from_lists([b"A", b"ACGT", b"NNNN", b"acgt", b"A" * 64, b"A" * 65, b"ACGTN" * 30],
                                [b"", b"ACGT", b"NNNN", b"ACGT", b"A" * 64, b"A" * 64, b"ACGTN" * 29])
    s = pyoracle.bpm(b)
    assert s[0] == -1 and s[1] == 0 and s[2] == 0 and s[3] == -4 and s[4] == 0 and s[5] == -1
