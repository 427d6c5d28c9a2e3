"""GPU parity: the bitpal HIP kernel (through the C ABI) vs the oracle and the golden scores of the reference."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_scores

pytestmark = pytest.mark.gpu
ALGS = {"bitpal_edit": 0, "bitpal_scored": 1}


@pytest.fixture(scope="module", params=[0, 1], ids=["edit", "scored"])
def eng(request):
    from genarchbench_amd.bitpal import BitpalEngine
    e = BitpalEngine(request.param)
    e.alg = request.param
    yield e
    e.close()


@pytest.mark.parametrize("alg", list(ALGS))
@pytest.mark.parametrize("name", ["bpm_bench", "bpm_adv"])
def test_golden(name, alg):
    from genarchbench_amd.bitpal import BitpalEngine
    batch = gabgen.read_pairs_text(f"{GOLDEN}/{name}.in.txt")
    want = read_scores(f"{GOLDEN}/{name}.{alg}.expected.txt")
    e = BitpalEngine(ALGS[alg])
    np.testing.assert_array_equal(e.benchmark_bitpal(batch), want)
    np.testing.assert_array_equal(e.benchmark_bitpal(batch.swapped_longer_first()), want)
    e.close()


@pytest.mark.parametrize("seed,n,mode,plen", [(41, 100000, 0, 151), (42, 20000, 1, 300), (43, 30000, 0, 100),
                                              (44, 2000, 1, 900), (45, 65, 1, 40), (46, 1, 0, 151), (47, 5000, 1, 33)])
def test_vs_oracle(eng, seed, n, mode, plen):
    batch = gabgen.pairs(seed, n, mode, plen)
    np.testing.assert_array_equal(eng.benchmark_bitpal(batch), pyoracle.bitpal(batch, eng.alg))
    st = eng.last_stats()
    assert st["cells"] == int((batch.pat_len.astype(np.int64) * batch.txt_len).sum()) and st["long_pairs"] == 0


def test_edge_lengths_and_bytes(eng):
    """empty strings, lengths around the 32-column chunk and the 4-byte loads, bytes outside ACGT (compared raw)"""
    rng = np.random.default_rng(5)
    pats, txts = [b"", b"A", b"", b"ACGT", b"acgt", b"NNNN", b"A" * 32, b"A" * 33, b"A" * 31, b"ACGT" * 16 + b"A", b"\xfe\xffAC"], \
                 [b"", b"", b"ACG", b"ACGT", b"ACGT", b"NNNN", b"A" * 32, b"A" * 32, b"C" * 64, b"ACGT" * 16, b"\xff\xfeAC"]
    for ln in list(range(1, 70)) + [95, 96, 97, 127, 128, 129]:
        p = rng.choice(np.frombuffer(b"ACGTN", np.uint8), ln).tobytes()
        t = bytearray(p)
        for _ in range(ln // 8):
            t[int(rng.integers(0, len(t)))] = b"ACGT"[int(rng.integers(0, 4))]
        cut = int(rng.integers(0, ln))
        pats.append(p); txts.append(bytes(t[:cut]) + bytes(t[min(ln, cut + int(rng.integers(0, 4))):]))
    b = gabgen.pairs_from_lists(pats, txts)
    np.testing.assert_array_equal(eng.benchmark_bitpal(b), pyoracle.bitpal(b, eng.alg))


def test_edit_bit_vector_path_and_its_rejects():
    """-a bitpal-edit runs Myers' bit-vector (rows = the shorter string, masks for A C G T N) and hands the integer DP what it
    cannot take: every row count from 1 to 300 (all word boundaries, beyond its 256 rows), bytes outside ACGTN in the row
    string (reject) and in the column string only (no reject), lower case, accepted and rejected pairs side by side in a wave"""
    from genarchbench_amd.bitpal import BitpalEngine
    rng = np.random.default_rng(17)
    alpha = np.frombuffer(b"ACGTN", np.uint8)
    pats, txts = [], []
    for ln in range(1, 301):
        p = bytearray(rng.choice(alpha, ln).tobytes())
        t = bytearray(p) + bytearray(rng.choice(alpha, int(rng.integers(0, 40))).tobytes())
        for _ in range(1 + ln // 10):
            t[int(rng.integers(0, len(t)))] = b"ACGT"[int(rng.integers(0, 4))]
        kind = ln % 5
        if kind == 1: t[int(rng.integers(0, len(t)))] = ord("x")            # column string only: stays on the bit-vector path
        if kind == 2: p[int(rng.integers(0, len(p)))] = ord("a")            # row string: integer DP
        if kind == 3: p, t = t, p                                          # the longer one first
        pats.append(bytes(p)); txts.append(bytes(t))
    b = gabgen.pairs_from_lists(pats * 8, txts * 8)
    e = BitpalEngine(0)
    np.testing.assert_array_equal(e.benchmark_bitpal(b), pyoracle.bitpal(b, 0))
    assert e.last_stats()["cells"] == int((b.pat_len.astype(np.int64) * b.txt_len).sum())
    e.close()


def test_edit_through_the_integer_dp(monkeypatch):
    """GAB_BITPAL_NO_BV=1: -a bitpal-edit through bitpal_dp<., false> alone (the kernel the bit-vector path's rejects take)"""
    from genarchbench_amd.bitpal import BitpalEngine
    monkeypatch.setenv("GAB_BITPAL_NO_BV", "1")
    e = BitpalEngine(0)
    for seed, n, mode, plen in ((48, 50000, 0, 151), (49, 10000, 1, 300)):
        batch = gabgen.pairs(seed, n, mode, plen)
        np.testing.assert_array_equal(e.benchmark_bitpal(batch), pyoracle.bitpal(batch, 0))
    e.close()


def test_long_pairs_global_path(eng):
    """row strings beyond the LDS column (2048 rows) -> int32 boundary column in global memory; mixed with short pairs"""
    rng = np.random.default_rng(8)
    pats, txts = [], []
    for n, err in ((1500, 0.05), (4000, 0.1), (2100, 0.3), (200, 0.1), (16320, 0.02), (3000, 0.0)):
        p = rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
        t = bytearray()
        for c in p:
            r = rng.random()
            if r < err / 3: continue
            if r < 2 * err / 3: t.append(b"ACGT"[int(rng.integers(0, 4))])
            t.append(c if r > err else b"ACGT"[int(rng.integers(0, 4))])
        pats.append(p); txts.append(bytes(t[:16320]))
    b = gabgen.pairs_from_lists(pats, txts)
    np.testing.assert_array_equal(eng.benchmark_bitpal(b), pyoracle.bitpal(b, eng.alg))
    assert eng.last_stats()["long_pairs"] == int((np.minimum(b.pat_len, b.txt_len) > 2048).sum()) >= 3


def test_limits_are_reported(eng):
    from genarchbench_amd._lib import GabError
    b = gabgen.pairs_from_lists([b"A" * 16321], [b"A"])
    with pytest.raises(GabError):
        eng.benchmark_bitpal(b)


def test_device_resident(eng):
    import torch
    batch = gabgen.pairs(48, 50000, 0, 151)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    sc = torch.zeros(batch.n, dtype=torch.int32, device=dev)
    eng.run_device(t(batch.pat), t(batch.pat_off), t(batch.pat_len), t(batch.txt), t(batch.txt_off), t(batch.txt_len), sc,
                   stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sc.cpu().numpy(), pyoracle.bitpal(batch, eng.alg))
