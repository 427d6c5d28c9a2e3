"""CPU: chain / fast-chain oracles (oracle/chain.c) against golden vectors from the compiled reference."""
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_chain_output


@pytest.mark.parametrize("name", ["chain_bench", "chain_dense"])
@pytest.mark.parametrize("mode,tag", [(0, "chain"), (1, "fastchain")])
def test_oracle_matches_golden(name, mode, tag):
    batch = gabgen.read_chain_text(f"{GOLDEN}/{name}.in.txt")
    ws, wp = read_chain_output(f"{GOLDEN}/{name}.{tag}.expected.txt")
    s, p = pyoracle.chain(batch, mode)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)


def test_dense_fixture_separates_the_two_modes():
    """the dense fixture must exercise max_skip: chain and fast-chain outputs differ on it"""
    a = read_chain_output(f"{GOLDEN}/chain_dense.chain.expected.txt")
    b = read_chain_output(f"{GOLDEN}/chain_dense.fastchain.expected.txt")
    assert (a[0] != b[0]).sum() + (a[1] != b[1]).sum() > 100


def test_generator_matches_text_fixture():
    a = gabgen.read_chain_text(f"{GOLDEN}/chain_dense.in.txt")
    b = gabgen.chain(202, 5, 1, 1500, 6000)
    np.testing.assert_array_equal(a.x, b.x); np.testing.assert_array_equal(a.y, b.y)
    assert (a.hdr == b.hdr).all()


def test_thread_invariant():
    b = gabgen.chain(9, 30, 1, 200, 3000)
    for mode in (0, 1):
        a1 = pyoracle.chain(b, mode, threads=1); a3 = pyoracle.chain(b, mode, threads=3)
        np.testing.assert_array_equal(a1[0], a3[0]); np.testing.assert_array_equal(a1[1], a3[1])


@pytest.mark.skipif(pyoracle.ref_path("chain_ref") is None, reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("mode,exe", [(0, "chain_ref"), (1, "fastchain_ref_avx2")])
def test_oracle_matches_live_reference(tmp_path, mode, exe):
    p = str(tmp_path / "in.txt"); o = str(tmp_path / "out.txt")
    gabgen.write_text("chain", p, 77, 12, 1, 800, 7000)
    subprocess.run([pyoracle.ref_path(exe), "-i", p, "-o", o, "-t", "2"], capture_output=True, check=True)
    ws, wp = read_chain_output(o)
    s, pa = pyoracle.chain(gabgen.chain(77, 12, 1, 800, 7000), mode)
    np.testing.assert_array_equal(s, ws); np.testing.assert_array_equal(pa, wp)
