"""CPU: the bsw oracle (oracle/bsw.c) against the golden vectors produced by the compiled reference,
and against the compiled reference run live when oracle/_ref is present."""
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_bsw_full, read_bsw_input, read_scores


@pytest.mark.parametrize("name", ["bsw_bench", "bsw_adv"])
def test_oracle_matches_golden(name):
    batch = read_bsw_input(f"{GOLDEN}/{name}.in.txt")
    want = read_scores(f"{GOLDEN}/{name}.expected.txt")
    got = pyoracle.bsw(batch)[:, 0]
    assert batch.n == len(want)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("name", ["bsw_bench", "bsw_adv"])
def test_oracle_full_result_matches_golden(name):
    """all six fields (score, qle, tle, gtle, gscore, max_off) against the reference's scalarBandedSWAWrapper ==
    getScores16 output (bsw/src/bandedSWA.cpp:241-252,258-276)"""
    batch = read_bsw_input(f"{GOLDEN}/{name}.in.txt")
    want = read_bsw_full(f"{GOLDEN}/{name}.full.expected.txt")
    assert want.shape == (batch.n, 6)
    np.testing.assert_array_equal(pyoracle.bsw(batch), want)
    np.testing.assert_array_equal(want[:, 0], read_scores(f"{GOLDEN}/{name}.expected.txt"))


@pytest.mark.skipif(pyoracle.ref_path("bsw_full_ref_avx2") is None, reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("how", ["scalar", "vector"])
def test_oracle_full_result_matches_live_reference(tmp_path, how):
    """fresh seed, adversarial mode: six fields straight against the reference's class"""
    n, seed = 4096, 993
    p = str(tmp_path / "in.txt")
    gabgen.write_text("bsw", p, seed, n, 1)
    r = subprocess.run([pyoracle.ref_path("bsw_full_ref_avx2"), p, how], capture_output=True, text=True, check=True)
    want = np.array([[int(v) for v in l.split()[1:]] for l in r.stdout.splitlines()], np.int32)
    np.testing.assert_array_equal(pyoracle.bsw(gabgen.bsw(seed, n, 1)), want)


@pytest.mark.parametrize("name,seed,n,mode", [("bsw_bench", 101, 2048, 0), ("bsw_adv", 102, 1536, 1)])
def test_generator_reproduces_golden_input(name, seed, n, mode):
    """the in-memory generator and the committed text fixture are the same data"""
    a = read_bsw_input(f"{GOLDEN}/{name}.in.txt")
    b = gabgen.bsw(seed, n, mode)
    np.testing.assert_array_equal(a.len1, b.len1)
    np.testing.assert_array_equal(a.len2, b.len2)
    np.testing.assert_array_equal(a.h0, b.h0)
    np.testing.assert_array_equal(a.ref[:a.ref_off[-1] + a.len1[-1]], b.ref[:b.ref_off[-1] + b.len1[-1]])
    np.testing.assert_array_equal(a.qry[:a.qry_off[-1] + a.len2[-1]], b.qry[:b.qry_off[-1] + b.len2[-1]])


def test_oracle_thread_invariant():
    b = gabgen.bsw(5, 4000, 1)
    np.testing.assert_array_equal(pyoracle.bsw(b, threads=1), pyoracle.bsw(b, threads=3))


def test_oracle_edge_cases():
    """hand-made pairs: 1-base sequences, all-N, no similarity, h0=0 (row max 0 -> immediate exit)"""
    A = lambda *x: np.array(x, np.uint8)
    refs = [A(0), A(1), A(4, 4, 4, 4), A(0, 1, 2, 3) , A(0, 0, 0, 0, 0, 0, 0, 0), A(0, 1, 2, 3)]
    qrys = [A(0), A(0), A(4, 4, 4, 4), A(0, 1, 2, 3), A(3, 3, 3, 3), A(0, 1, 2, 3)]
    h0 = [10, 10, 50, 0, 3, 1]
    out = pyoracle.bsw(gabgen.bsw_from_arrays(refs, qrys, h0))
    # scores: match extends h0 by 1; mismatch leaves the seed score; h0=0 can never extend
    assert out[0, 0] == 11 and out[1, 0] == 10 and out[2, 0] == 50 and out[3, 0] == 0
    assert out[4, 0] == 3 and out[5, 0] == 5


@pytest.mark.skipif(pyoracle.ref_path("bsw_ref_avx2") is None, reason="oracle/_ref not built (no /root/reference)")
def test_oracle_matches_live_reference(tmp_path):
    """fresh seed, adversarial mode, straight against the compiled reference"""
    n, seed = 4096, 991
    p = str(tmp_path / "in.txt")
    gabgen.write_text("bsw", p, seed, n, 1)
    r = subprocess.run([pyoracle.ref_path("bsw_ref_avx2"), "-pairs", p, "-t", "2", "-b", "256"],
                       capture_output=True, text=True, check=True)
    want = np.array([int(l.split("=")[1]) for l in r.stderr.splitlines() if "score=" in l][:n], np.int32)
    got = pyoracle.bsw(gabgen.bsw(seed, n, 1))[:, 0]
    np.testing.assert_array_equal(got, want)
