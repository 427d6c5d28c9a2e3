"""CPU: the bitpal oracle (oracle/bitpal.c) against golden scores from the compiled reference (-a bitpal-edit / bitpal-scored)."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_scores

ALGS = {"bitpal_edit": 0, "bitpal_scored": 1}


@pytest.mark.parametrize("alg", list(ALGS))
@pytest.mark.parametrize("name", ["bpm_bench", "bpm_adv"])
def test_oracle_matches_golden(name, alg):
    batch = gabgen.read_pairs_text(f"{GOLDEN}/{name}.in.txt")
    want = read_scores(f"{GOLDEN}/{name}.{alg}.expected.txt")
    np.testing.assert_array_equal(pyoracle.bitpal(batch, ALGS[alg]), want)
    # the score is symmetric in the two strings: the driver's longer-first swap does not matter
    np.testing.assert_array_equal(pyoracle.bitpal(batch.swapped_longer_first(), ALGS[alg]), want)


def test_edit_score_is_minus_levenshtein_and_bounds():
    b = gabgen.pairs(31, 400, 1, 90)
    ed = pyoracle.bitpal(b, 0); sc = pyoracle.bitpal(b, 1)
    for i in range(0, b.n, 5):
        p = bytes(b.pat[b.pat_off[i]:b.pat_off[i] + b.pat_len[i]]); t = bytes(b.txt[b.txt_off[i]:b.txt_off[i] + b.txt_len[i]])
        prev = list(range(len(t) + 1))
        for x, ca in enumerate(p, 1):
            cur = [x]
            for y, cb in enumerate(t, 1):
                cur.append(min(prev[y] + 1, cur[y - 1] + 1, prev[y - 1] + (ca != cb)))
            prev = cur
        assert ed[i] == -prev[-1]
    # scored: at most one point per base of the shorter string, never below the all-gap alignment
    ln = np.minimum(b.pat_len, b.txt_len); tot = b.pat_len + b.txt_len
    assert (sc <= ln - 2 * np.abs(b.pat_len - b.txt_len)).all() and (sc >= -2 * tot).all()
