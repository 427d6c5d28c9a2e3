"""GPU parity: the bsw HIP path (through the C ABI) against the CPU oracle and the golden vectors."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_bsw_full, read_bsw_input, read_scores

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sw():
    from genarchbench_amd.bsw import BandedPairWiseSW
    s = BandedPairWiseSW()
    yield s
    s.close()


@pytest.mark.parametrize("name", ["bsw_bench", "bsw_adv"])
def test_golden(sw, name):
    batch = read_bsw_input(f"{GOLDEN}/{name}.in.txt")
    want = read_scores(f"{GOLDEN}/{name}.expected.txt")
    np.testing.assert_array_equal(sw.getScores16(batch), want)


@pytest.mark.parametrize("seed,n,mode", [(11, 100000, 0), (12, 50000, 1), (13, 63, 0), (14, 65, 1), (15, 1, 0)])
def test_vs_oracle(sw, seed, n, mode):
    batch = gabgen.bsw(seed, n, mode)
    want = pyoracle.bsw(batch)[:, 0]
    np.testing.assert_array_equal(sw.getScores16(batch), want)


def test_full_result_and_cells(sw):
    """all six result fields + the DP cell counter, device-resident entry point"""
    import torch
    batch = gabgen.bsw(21, 20000, 1)
    want, cells = pyoracle.bsw(batch, want_cells=True)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    ref, qry = t(batch.ref), t(batch.qry)
    score = torch.full((batch.n,), -7, dtype=torch.int32, device=dev)
    res = torch.full((batch.n, 6), -7, dtype=torch.int32, device=dev)
    sw.run_device(ref, t(batch.ref_off), qry, t(batch.qry_off), t(batch.len1), t(batch.len2), t(batch.h0),
                  score, res, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(res.cpu().numpy(), want)
    np.testing.assert_array_equal(score.cpu().numpy(), want[:, 0])
    assert sw.last_stats()["cells"] == cells


@pytest.mark.parametrize("name", ["bsw_bench", "bsw_adv"])
def test_full_result_golden(sw, name):
    """result_out (all six fields) against the reference's own scalarBandedSWAWrapper / getScores16 output"""
    import torch
    batch = read_bsw_input(f"{GOLDEN}/{name}.in.txt")
    want = read_bsw_full(f"{GOLDEN}/{name}.full.expected.txt")
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    score = torch.full((batch.n,), -7, dtype=torch.int32, device=dev)
    res = torch.full((batch.n, 6), -7, dtype=torch.int32, device=dev)
    sw.run_device(t(batch.ref), t(batch.ref_off), t(batch.qry), t(batch.qry_off), t(batch.len1), t(batch.len2), t(batch.h0),
                  score, res, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(res.cpu().numpy(), want)


def test_wide_mode_large_h0(sw):
    """h0 beyond the 15-bit packed range switches to the 32-bit cell kernel"""
    batch = gabgen.bsw(31, 3000, 1)
    batch.h0[::7] = 40000
    batch.h0[1::7] = 32767 - 200
    want = pyoracle.bsw(batch)[:, 0]
    np.testing.assert_array_equal(sw.getScores16(batch), want)


def test_max_lengths(sw):
    """query 256 (the LDS-heaviest class) and a long reference"""
    rng = np.random.default_rng(3)
    qrys = [rng.integers(0, 4, 256).astype(np.uint8) for _ in range(70)]
    refs = []
    for q in qrys:
        r = q.copy()
        r[rng.integers(0, 256, 10)] = rng.integers(0, 4, 10)
        refs.append(np.concatenate([r, rng.integers(0, 4, rng.integers(0, 1700)).astype(np.uint8)]))
    batch = gabgen.bsw_from_arrays(refs, qrys, [int(x) for x in rng.integers(0, 300, 70)])
    want = pyoracle.bsw(batch)[:, 0]
    np.testing.assert_array_equal(sw.getScores16(batch), want)


def test_other_penalties():
    from genarchbench_amd.bsw import BandedPairWiseSW, bwa_fill_scmat
    batch = gabgen.bsw(41, 5000, 1)
    # (the byte-cell kernel is compiled per o_del + e_del == o_ins + e_ins and per "no score above 1": all four meet here)
    for (a, b, go, ge, amb, zd, w, d_ins) in [(2, 3, 5, 2, -2, 50, 30, 1), (1, 1, 0, 1, 0, 0, 100, 1), (3, 5, 7, 3, -1, 200, 5, 1),
                                             (2, 3, 5, 2, -2, 50, 30, 0), (1, 4, 6, 1, -1, 100, 100, 0)]:
        s = BandedPairWiseSW(go, ge, go + d_ins, ge, zd, 5, bwa_fill_scmat(a, b, amb), w)
        p = pyoracle.bsw_params(a, b, go, ge, amb, zd, 5, w)
        p.o_ins = go + d_ins
        np.testing.assert_array_equal(s.getScores16(batch), pyoracle.bsw(batch, p)[:, 0])
        s.close()


def test_host_window_from_a_sample_of_the_pairs(sw):
    """gab_bsw_run stages only the window of the slabs a batch uses and takes that window from 66 pairs spread over the batch
    (no serial scan of all offsets inside the ROI); the device checks every pair against it.  Sequences that do NOT lie in
    pair order fall outside the sampled window: the call then scans all pairs and stages again -- same scores; an invalid
    pair is still an error after that; batches in pair order keep working afterwards."""
    from genarchbench_amd._lib import GabError
    b = gabgen.bsw(21, 20000, 1)
    want = pyoracle.bsw(b)[:, 0]
    np.testing.assert_array_equal(sw.getScores16(b), want)                  # pair order: the sample is exact
    rng = np.random.default_rng(3)
    perm = rng.permutation(b.n)                                             # same slabs, pairs handed over in shuffled order,
    sh = gabgen.BswBatch(b.ref, b.ref_off[perm].copy(), b.qry, b.qry_off[perm].copy(), b.len1[perm].copy(), b.len2[perm].copy(), b.h0[perm].copy())
    lo, hi = 7000, 15000                                                    # ... and only a window of them: the extremes are inside
    sub = gabgen.BswBatch(sh.ref, sh.ref_off[lo:hi], sh.qry, sh.qry_off[lo:hi], sh.len1[lo:hi], sh.len2[lo:hi], sh.h0[lo:hi])
    np.testing.assert_array_equal(sw.getScores16(sub), want[perm][lo:hi])
    bad = gabgen.BswBatch(b.ref, b.ref_off, b.qry, b.qry_off, b.len1, b.len2.copy(), b.h0)
    bad.len2[12345] = 0
    with pytest.raises(GabError):
        sw.getScores16(bad)
    np.testing.assert_array_equal(sw.getScores16(b), want)


def test_host_window_pair_ending_in_the_padding_of_the_sampled_window():
    """the sampled window is padded to 256 bytes on the device, and only the sampled extent is copied.  A pair
    that was NOT sampled and ends less than 256 bytes past the sampled end used to pass the device's check against the padded
    window and read staging bytes nobody had copied (here: what the batch before left there).  The device now checks against
    the copied extent, so such a batch is scanned in full and staged again."""
    from genarchbench_amd.bsw import BandedPairWiseSW
    sw = BandedPairWiseSW(device=0)
    try:
        a = gabgen.bsw(77, 20000, 1)
        np.testing.assert_array_equal(sw.getScores16(a), pyoracle.bsw(a)[:, 0])          # fills the staging buffer with other data
        b = gabgen.bsw(22, 20000, 1)
        want = pyoracle.bsw(b)[:, 0]
        # the last two pairs handed over in swapped order: pair n - 1 (sampled) is now the one that lies first in the slabs,
        # pair n - 2 (not sampled: (n - 1) * 64 // 65 != n - 2) ends behind the sampled end
        n = b.n
        assert (n - 1) * 64 // 65 != n - 2
        idx = np.arange(n); idx[n - 2], idx[n - 1] = n - 1, n - 2
        sw2 = gabgen.BswBatch(b.ref, b.ref_off[idx].copy(), b.qry, b.qry_off[idx].copy(), b.len1[idx].copy(), b.len2[idx].copy(), b.h0[idx].copy())
        assert sw2.ref_off[n - 2] + sw2.len1[n - 2] > sw2.ref_off[n - 1] + sw2.len1[n - 1]
        assert sw2.ref_off[n - 2] + sw2.len1[n - 2] - (sw2.ref_off[n - 1] + sw2.len1[n - 1]) < 256
        np.testing.assert_array_equal(sw.getScores16(sw2), want[idx])
        np.testing.assert_array_equal(sw.getScores16(b), want)
    finally:
        sw.close()


def test_rejects_bad_input(sw):
    from genarchbench_amd._lib import GabError
    A = lambda *x: np.array(x, np.uint8)
    batch = gabgen.bsw_from_arrays([A(0, 1)], [np.zeros(257, np.uint8)], [5])
    with pytest.raises(GabError):
        sw.getScores16(batch)
    empty = gabgen.bsw_from_arrays([], [], [])
    assert len(sw.getScores16(empty)) == 0
    # device entry point: a sequence needs three more bytes behind it INSIDE its slab (the kernels read dwords from the sequence's
    # own start): a slab that ends with the last base is refused, with three bytes of slack it is taken
    import torch
    ok = gabgen.bsw_from_arrays([A(0, 1, 2, 3, 0), A(1, 1, 2)], [A(0, 1, 2, 3, 0), A(1, 1, 2)], [5, 5])
    want = pyoracle.bsw(ok)[:, 0]
    total = int(ok.ref_off[-1] + ok.len1[-1])
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for slack, good in ((0, False), (2, False), (3, True)):
        score = torch.full((ok.n,), -7, dtype=torch.int32, device=dev)
        args = (t(ok.ref[:total + slack]), t(ok.ref_off), t(ok.qry), t(ok.qry_off), t(ok.len1), t(ok.len2), t(ok.h0), score, None)
        if good:
            sw.run_device(*args, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(score.cpu().numpy(), want)
        else:
            with pytest.raises(GabError):
                sw.run_device(*args, stream=torch.cuda.current_stream().cuda_stream)


def test_knobs_are_read_when_the_handle_is_made(monkeypatch, capfd):
    """the GAB_* experiment switches are read ONCE per handle (gab_internal.h: gab_tuning): a switch set after the
    handle exists changes nothing -- unless GAB_TUNING_LIVE is set, as the test suite does for its own handles"""
    from genarchbench_amd.bsw import BandedPairWiseSW
    b = gabgen.bsw(5, 300)
    monkeypatch.delenv("GAB_TUNING_LIVE", raising=False)
    monkeypatch.delenv("GAB_BSW_TRACE", raising=False)
    e = BandedPairWiseSW()
    monkeypatch.setenv("GAB_BSW_TRACE", "1")                    # after the handle was made: not seen
    want = e.getScores16(b)
    assert "[gab_bsw_run" not in capfd.readouterr().err
    e2 = BandedPairWiseSW()                                     # a handle made now has it
    np.testing.assert_array_equal(e2.getScores16(b), want)
    assert "[gab_bsw_run" in capfd.readouterr().err
    monkeypatch.setenv("GAB_TUNING_LIVE", "1")                  # the tests' mode: every call reads the switches again
    np.testing.assert_array_equal(e.getScores16(b), want)
    assert "[gab_bsw_run" in capfd.readouterr().err
    monkeypatch.delenv("GAB_BSW_TRACE")
    np.testing.assert_array_equal(e.getScores16(b), want)
    assert "[gab_bsw_run" not in capfd.readouterr().err
    e.close(); e2.close()
