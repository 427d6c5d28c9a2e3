"""GPU parity: bpm HIP kernels (through the C ABI) vs the oracle and the golden vectors."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_scores

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from genarchbench_amd.bpm import BpmEngine
    e = BpmEngine()
    yield e
    e.close()


@pytest.mark.parametrize("name", ["bpm_bench", "bpm_adv"])
def test_golden(eng, name):
    batch = gabgen.read_pairs_text(f"{GOLDEN}/{name}.in.txt").swapped_longer_first()
    want = read_scores(f"{GOLDEN}/{name}.expected.txt")
    np.testing.assert_array_equal(eng.benchmark_edit_bpm(batch), want)


@pytest.mark.parametrize("seed,n,mode,plen", [(51, 200000, 0, 151), (52, 60000, 1, 256), (53, 30000, 0, 100),
                                              (54, 3000, 1, 700), (55, 65, 1, 64), (56, 1, 0, 151)])
def test_vs_oracle(eng, seed, n, mode, plen):
    batch = gabgen.pairs(seed, n, mode, plen).swapped_longer_first()
    want, steps = pyoracle.bpm(batch, want_steps=True)
    got = eng.benchmark_edit_bpm(batch)
    np.testing.assert_array_equal(got, want)
    st = eng.last_stats()
    assert st["block_steps"] >= steps          # queued pairs are stepped twice (score pass + full pass)


def test_block_form_score_kernel(eng, monkeypatch):
    """GAB_BPM_SCORE64=1: the score pass through bpm_score<W>, the 64-row block form of BPM_ADVANCE_BLOCK (the default is the
    D-word form, bpm_score32<D>); mode 1 of the generator mixes pattern lengths, i.e. odd and even word counts, in one batch"""
    monkeypatch.setenv("GAB_BPM_SCORE64", "1")
    for seed, n, mode, plen in ((57, 100000, 0, 151), (58, 40000, 1, 256)):
        batch = gabgen.pairs(seed, n, mode, plen).swapped_longer_first()
        np.testing.assert_array_equal(eng.benchmark_edit_bpm(batch), pyoracle.bpm(batch))


@pytest.mark.parametrize("plen", [32, 33, 64, 65, 96, 97, 128, 129, 160, 161, 192, 193, 224, 225, 256])
def test_word_count_boundaries(eng, plen):
    """pattern lengths on both sides of every 32-row word boundary: the D-word kernels pick D = 2W - 1 or 2W by the longest
    pattern of the class, and the distance bit moves between the last two words"""
    batch = gabgen.pairs(600 + plen, 4000, 0, plen).swapped_longer_first()
    np.testing.assert_array_equal(eng.benchmark_edit_bpm(batch), pyoracle.bpm(batch))


def test_sliced_classes(eng):
    """classes of >= 2^18 pairs are scored in slices whose band kernels run on a second stream (device-side queue
    lengths): two populated classes, unclean pairs in every slice"""
    a = gabgen.pairs(58, 600000, 0, 151)
    b = gabgen.pairs(59, 700000, 1, 100)
    pats = np.concatenate([a.pat, b.pat]); txts = np.concatenate([a.txt, b.txt])
    batch = gabgen.PairBatch(pats, np.concatenate([a.pat_off, b.pat_off + len(a.pat)]), np.concatenate([a.pat_len, b.pat_len]),
                             txts, np.concatenate([a.txt_off, b.txt_off + len(a.txt)]), np.concatenate([a.txt_len, b.txt_len])).swapped_longer_first()
    np.testing.assert_array_equal(eng.benchmark_edit_bpm(batch), pyoracle.bpm(batch))
    assert eng.last_stats()["full_pairs"] > 10000


def test_edge_cases(eng):
    pats = [b"A", b"ACGT", b"NNNN", b"acgt", b"A" * 64, b"A" * 65, b"ACGTN" * 30, b"N" * 130, b"ACGT" * 64, b"T" * 257]
    txts = [b"", b"ACGT", b"NNNN", b"ACGT", b"A" * 64, b"A" * 64, b"ACGTN" * 29, b"A" * 129, b"ACGT" * 63 + b"ACG", b"T" * 200]
    b = gabgen.pairs_from_lists(pats, txts)
    np.testing.assert_array_equal(eng.benchmark_edit_bpm(b), pyoracle.bpm(b))


def test_long_patterns_generic_path(eng):
    """W > 4 words: the generic history kernel; includes the reference maximum of 255 words"""
    rng = np.random.default_rng(9)
    pats, txts = [], []
    for n in (257, 300, 1000, 5000, 16320):
        p = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n, p=[.24, .24, .24, .24, .04]).tobytes()
        t = bytearray(p[: n - int(rng.integers(0, 40))])
        for k in rng.integers(0, len(t), max(1, len(t) // 50)):
            t[k] = b"ACGT"[int(rng.integers(0, 4))]
        pats.append(p); txts.append(bytes(t))
    b = gabgen.pairs_from_lists(pats, txts)
    np.testing.assert_array_equal(eng.benchmark_edit_bpm(b), pyoracle.bpm(b))


def test_rejects_text_longer_than_pattern(eng):
    from genarchbench_amd._lib import GabError
    b = gabgen.pairs_from_lists([b"ACG"], [b"ACGTACGT"])
    with pytest.raises(GabError):
        eng.benchmark_edit_bpm(b)


def test_device_resident(eng):
    import torch
    batch = gabgen.pairs(57, 50000, 0, 151).swapped_longer_first()
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    sc = torch.zeros(batch.n, dtype=torch.int32, device=dev)
    eng.run_device(t(batch.pat), t(batch.pat_off), t(batch.pat_len), t(batch.txt), t(batch.txt_off), t(batch.txt_len),
                   sc, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sc.cpu().numpy(), pyoracle.bpm(batch))


def test_many_long_patterns_generic_path(eng):
    """every pair above 256 bases takes the generic (W > 4) path: their lengths come back in ONE copy (a gather kernel),
    not two 4-byte copies per pair"""
    rng = np.random.default_rng(17)
    pats, txts = [], []
    for _ in range(3000):
        n = int(rng.integers(257, 700))
        p = rng.integers(0, 4, n)
        t = p.copy()
        t[rng.integers(0, n, 12)] = rng.integers(0, 4, 12)
        t = np.delete(t, rng.integers(0, n, 3))
        lut = np.frombuffer(b"ACGT", np.uint8)
        pats.append(lut[p].tobytes()); txts.append(lut[t].tobytes())
    b = gabgen.pairs_from_lists(pats, txts)
    np.testing.assert_array_equal(eng.benchmark_edit_bpm(b), pyoracle.bpm(b))
    assert eng.last_stats()["full_pairs"] >= 3000
