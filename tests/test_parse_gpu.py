"""GPU parity: the input parsers against the line-by-line readers of tests/util.py /
tools/gabgen.py (restatements of the reference drivers' parsers), and end to end against the reference's golden output."""
import ctypes as C

import numpy as np
import pytest

from tools import gabgen
from tests.util import GOLDEN, read_bsw_input, read_scores

pytestmark = pytest.mark.gpu


def _bsw_text(seed, n, mode=0, tmp=None):
    p = str(tmp / "in.txt")
    gabgen.write_text("bsw", p, seed, n, mode)
    return p, open(p, "rb").read()


@pytest.mark.parametrize("name", ["bsw_bench", "bsw_adv"])
def test_bsw_parse_golden_end_to_end(name):
    """GPU parse -> GPU banded SW == the reference's expected scores; parsed arrays == the line-by-line reader"""
    import torch
    from genarchbench_amd.parse import InputParser
    from genarchbench_amd.bsw import BandedPairWiseSW
    text = open(f"{GOLDEN}/{name}.in.txt", "rb").read()
    ps = InputParser()
    pk = ps.bsw_pairs(text)
    want = read_bsw_input(f"{GOLDEN}/{name}.in.txt")
    got = ps.bsw_to_host(pk)
    assert pk.n == len(want.len1)
    np.testing.assert_array_equal(got["len1"], want.len1); np.testing.assert_array_equal(got["len2"], want.len2)
    np.testing.assert_array_equal(got["h0"], want.h0)
    for i in range(0, pk.n, 7):
        np.testing.assert_array_equal(got["ref"][got["ref_off"][i]:got["ref_off"][i] + got["len1"][i]],
                                      want.ref[want.ref_off[i]:want.ref_off[i] + want.len1[i]])
        np.testing.assert_array_equal(got["qry"][got["qry_off"][i]:got["qry_off"][i] + got["len2"][i]],
                                      want.qry[want.qry_off[i]:want.qry_off[i] + want.len2[i]])
    # straight into the DP through the device entry point
    sw = BandedPairWiseSW(device=0)
    score = torch.empty(pk.n, dtype=torch.int32, device="cuda:0")
    from genarchbench_amd._lib import check, lib
    check(lib().gab_bsw_run_device(sw._h, C.c_void_p(pk.d_ref), C.c_int64(pk.ref_bytes), C.c_void_p(pk.d_ref_off),
                                   C.c_void_p(pk.d_qry), C.c_int64(pk.qry_bytes), C.c_void_p(pk.d_qry_off), C.c_void_p(pk.d_len1),
                                   C.c_void_p(pk.d_len2), C.c_void_p(pk.d_h0), C.c_int64(pk.n), C.c_void_p(score.data_ptr()),
                                   C.c_void_p(0), C.c_void_p(0)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(score.cpu().numpy(), read_scores(f"{GOLDEN}/{name}.expected.txt")[:pk.n])
    sw.close(); ps.close()


def test_bsw_parse_large_and_block_boundaries(tmp_path):
    """200 k pairs (~40 MB: thousands of 16 KB blocks, lines straddling every kind of boundary) vs the seeded generator"""
    from genarchbench_amd.parse import InputParser
    n = 200_000
    path, text = _bsw_text(31, n, 0, tmp_path)
    want = gabgen.bsw(31, n, 0)
    ps = InputParser()
    pk = ps.bsw_pairs(text)
    got = ps.bsw_to_host(pk)
    assert pk.n == n
    np.testing.assert_array_equal(got["len1"], want.len1); np.testing.assert_array_equal(got["len2"], want.len2)
    np.testing.assert_array_equal(got["h0"], want.h0)
    # every sequence starts on a 4-byte boundary right after the previous one
    np.testing.assert_array_equal(got["ref_off"], np.concatenate([[0], np.cumsum((want.len1[:-1] + 3) & ~3, dtype=np.int64)]))
    np.testing.assert_array_equal(got["qry_off"], np.concatenate([[0], np.cumsum((want.len2[:-1] + 3) & ~3, dtype=np.int64)]))
    for i in list(range(0, 3000)) + list(range(3000, n, 997)):
        np.testing.assert_array_equal(got["ref"][got["ref_off"][i]:got["ref_off"][i] + got["len1"][i]],
                                      want.ref[want.ref_off[i]:want.ref_off[i] + want.len1[i]])
        np.testing.assert_array_equal(got["qry"][got["qry_off"][i]:got["qry_off"][i] + got["len2"][i]],
                                      want.qry[want.qry_off[i]:want.qry_off[i] + want.len2[i]])
    tail = n - 1
    np.testing.assert_array_equal(got["qry"][got["qry_off"][tail]:got["qry_off"][tail] + got["len2"][tail]],
                                  want.qry[want.qry_off[tail]:want.qry_off[tail] + want.len2[tail]])
    assert ps.last_stats()["kernel_ms"] > 0
    ps.close()


def test_bsw_parse_rejects_what_the_reference_cannot_read():
    from genarchbench_amd._lib import GabError
    from genarchbench_amd.parse import InputParser
    ps = InputParser()
    ok = b"19\n0123\n012\n"
    assert ps.bsw_pairs(ok).n == 1
    assert ps.bsw_pairs(ok + b"7\n01\n").n == 1            # incomplete trailing pair: newline count / 3
    assert ps.bsw_pairs(b"").n == 0
    got = ps.bsw_to_host(ps.bsw_pairs(b" +42\n3210\n01\n-7\n4\n/5\n"))
    assert list(got["h0"]) == [42, -7] and list(got["len1"]) == [4, 1] and list(got["ref"][:5]) == [3, 2, 1, 0, 4]
    assert list(got["qry"][got["qry_off"][1]:got["qry_off"][1] + 2]) == [255, 5]      # '/' - '0' wraps like the reference's uint8
    for bad in (b"19\n\n012\n", b"19\n0123\n\n", b"123456789\n01\n01\n", b"1\n" + b"0" * 2046 + b"\n01\n", b"1\n01\n" + b"1" * 254 + b"\n"):
        with pytest.raises(GabError):
            ps.bsw_pairs(bad)
    ps.close()


@pytest.mark.parametrize("name,swap", [("bpm_bench", True), ("bpm_adv", True), ("wfa_bench", False), ("wfa_adv", False)])
def test_pairs_parse_golden(name, swap):
    from genarchbench_amd.parse import InputParser
    path = f"{GOLDEN}/{name}.in.txt"
    text = open(path, "rb").read()
    ps = InputParser()
    pk = ps.pairs(text, swap)
    got = ps.pairs_to_host(pk)
    lines = text.split(b"\n")
    n = len(lines) // 2
    assert pk.n == n
    buf = np.frombuffer(text, np.uint8)
    for i in range(n):
        a, b = lines[2 * i][1:], lines[2 * i + 1][1:]
        if swap and len(b) > len(a):
            a, b = b, a
        assert got["pat_len"][i] == len(a) and got["txt_len"][i] == len(b)
        assert bytes(buf[got["pat_off"][i]:got["pat_off"][i] + len(a)]) == a
        assert bytes(buf[got["txt_off"][i]:got["txt_off"][i] + len(b)]) == b
    ps.close()


def test_pairs_parse_feeds_bpm_and_wfa():
    """the parsed device buffers go straight into gab_bpm_run_device / gab_wfa_run_device and reproduce the golden output"""
    import torch
    from genarchbench_amd._lib import check, lib
    from genarchbench_amd.parse import InputParser
    from genarchbench_amd.bpm import BpmEngine
    text = open(f"{GOLDEN}/bpm_bench.in.txt", "rb").read()
    ps = InputParser()
    pk = ps.pairs(text, True)
    be = BpmEngine(device=0)
    score = torch.empty(pk.n, dtype=torch.int32, device="cuda:0")
    check(lib().gab_bpm_run_device(be._h, C.c_void_p(pk.d_text), C.c_int64(pk.text_bytes), C.c_void_p(pk.d_pat_off), C.c_void_p(pk.d_pat_len),
                                   C.c_void_p(pk.d_text), C.c_int64(pk.text_bytes), C.c_void_p(pk.d_txt_off), C.c_void_p(pk.d_txt_len),
                                   C.c_int64(pk.n), C.c_void_p(score.data_ptr()), C.c_void_p(0)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(score.cpu().numpy(), read_scores(f"{GOLDEN}/bpm_bench.expected.txt"))
    be.close(); ps.close()


@pytest.mark.parametrize("name", ["chain_bench", "chain_dense"])
def test_chain_parse_golden_end_to_end(name):
    """GPU parse == the token reader; GPU parse -> GPU chain / fast-chain == the reference's expected output"""
    import torch
    from genarchbench_amd._lib import check, lib
    from genarchbench_amd.parse import InputParser
    from genarchbench_amd.chain import ChainEngine
    from tests.util import read_chain_output
    path = f"{GOLDEN}/{name}.in.txt"
    text = open(path, "rb").read()
    want = gabgen.read_chain_text(path)
    ps = InputParser()
    pk = ps.chain(text)
    got = ps.chain_to_host(pk)
    assert pk.ncalls == want.ncalls and pk.total == want.nanchors
    np.testing.assert_array_equal(got["call_off"][:-1], want.call_off)
    for f in ("n", "avg_qspan", "max_dist_x", "max_dist_y", "bw", "n_segs"):
        np.testing.assert_array_equal(got["hdr"][f], want.hdr[f], err_msg=f)
    np.testing.assert_array_equal(got["x"], want.x); np.testing.assert_array_equal(got["y"], want.y)
    ce = ChainEngine(device=0)
    for mode, tag in ((0, "chain"), (1, "fastchain")):
        score = torch.empty(pk.total, dtype=torch.int32, device="cuda:0"); parent = torch.empty_like(score)
        check(lib().gab_chain_run_device(ce._h, C.c_int(mode), C.c_void_p(pk.d_x), C.c_void_p(pk.d_y), pk.call_off, C.c_void_p(pk.hdr),
                                         C.c_int64(pk.ncalls), C.c_void_p(score.data_ptr()), C.c_void_p(parent.data_ptr()), C.c_void_p(0)))
        torch.cuda.synchronize()
        ws, wp = read_chain_output(f"{GOLDEN}/{name}.{tag}.expected.txt")
        np.testing.assert_array_equal(score.cpu().numpy(), ws); np.testing.assert_array_equal(parent.cpu().numpy(), wp)
    ce.close(); ps.close()


def test_chain_parse_declines_other_layouts():
    from genarchbench_amd._lib import GabError
    from genarchbench_amd.parse import InputParser
    ps = InputParser()
    ok = b"2\t15.000000\t5000\t5000\t500\t1\n10\t20\n30\t40\nEOR\n"
    pk = ps.chain(ok)
    got = ps.chain_to_host(pk)
    assert pk.ncalls == 1 and list(got["x"]) == [10, 30] and list(got["y"]) == [20, 40] and got["hdr"]["avg_qspan"][0] == np.float32(15.0)
    two = ps.chain(ok + b"0\t1.5\t1\t2\t3\t4\nEOR\n")
    assert two.ncalls == 2 and two.total == 2
    for bad in (b"3\t15.0\t5000\t5000\t500\t1\n10\t20\n30\t40\nEOR\n",            # header n != anchor lines
                b"2\t15.0\t5000\t5000\t500\t1\n10 20 30 40\nEOR\n",              # all anchors on one line (legal for fscanf)
                b"1\t15.0\t5000\t5000\t500\t1\n-5\t20\nEOR\n",                    # sign
                b"1\t15.0\t5000\t5000\t500\t1\n99999999999999999999\t20\nEOR\n",  # overflows 64 bits
                b"1\t15.0\t5000\t5000\t500\t1\n10\t20\n"):                         # no EOR
        with pytest.raises(GabError):
            ps.chain(bad)
    ps.close()
