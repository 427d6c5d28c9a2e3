"""GPU parity: wfa HIP kernels (through the C ABI) vs the oracle and the golden CIGARs."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_cigars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from genarchbench_amd.wfa import AffineWavefronts
    e = AffineWavefronts()
    yield e
    e.close()


def same(res_gpu, res_cpu):
    go, goff, gl, gs = res_gpu[:4]
    co, coff, cl, cs = res_cpu[:4]
    np.testing.assert_array_equal(gs, cs)
    np.testing.assert_array_equal(gl, cl)
    np.testing.assert_array_equal(goff, coff)
    # ops slabs: compare only the bytes each pair owns
    mask = np.zeros(len(co), bool)
    for o, l in zip(coff, cl):
        mask[o:o + l] = True
    np.testing.assert_array_equal(go[:len(co)][mask], co[mask])


@pytest.mark.parametrize("name", ["wfa_bench", "wfa_adv"])
def test_golden(eng, name):
    batch = gabgen.read_pairs_text(f"{GOLDEN}/{name}.in.txt")
    want = read_cigars(f"{GOLDEN}/{name}.expected.txt")
    assert pyoracle.wfa_cigars(eng.align(batch)) == want


@pytest.mark.parametrize("seed,n,mode,plen", [(61, 100000, 0, 151), (62, 20000, 1, 300), (63, 20000, 0, 100),
                                              (64, 3000, 1, 900), (65, 65, 1, 40), (66, 1, 0, 151)])
def test_vs_oracle(eng, seed, n, mode, plen):
    batch = gabgen.pairs(seed, n, mode, plen)
    want = pyoracle.wfa(batch, want_cells=True)
    got = eng.align(batch)
    same(got, want)
    assert eng.last_stats()["work"] == want[4]


def test_edge_and_padding_chars(eng):
    """empty strings, X/Y bytes that collide with the reference's padding characters, long indels"""
    pats = [b"A", b"ACGT", b"", b"ACGT", b"AAAA", b"ACGTACGT", b"XXYY", b"ACGTXX", b"YYACGT", b"A" * 200, b"ACGT" * 50]
    txts = [b"A", b"", b"ACGT", b"AGGT", b"AAAAAAAA", b"ACGT", b"YYXX", b"ACGTXXXXXX", b"ACGT", b"A" * 120, b"TGCA" * 50]
    b = gabgen.pairs_from_lists(pats, txts)
    same(eng.align(b), pyoracle.wfa(b))


def test_padding_characters_inside_the_strings(eng):
    """bases that ARE the reference's padding bytes: a pattern position beyond its end reads 'X', a text position beyond its
    end reads 'Y', so an 'X' in the text (a 'Y' in the pattern) can match the other string's padding and carry an extension
    past the end -- in every tier, at low and high scores"""
    rng = np.random.default_rng(21)
    alpha = np.frombuffer(b"ACGTXY", np.uint8)
    pats, txts = [], []
    for _ in range(4000):
        n = int(rng.integers(5, 150))
        err = float(rng.choice([0.0, 0.02, 0.05, 0.2]))
        p = rng.choice(alpha, n, p=[.2, .2, .2, .2, .1, .1]).tobytes()
        t = bytearray()
        for c in p:
            r = rng.random()
            if r < err / 3: continue
            if r < 2 * err / 3: t.append(int(rng.choice(alpha)))
            t.append(c if r > err else int(rng.choice(alpha)))
        # tails made of the other string's padding byte
        if rng.random() < 0.3: t += b"X" * int(rng.integers(1, 12))
        if rng.random() < 0.3: p += b"Y" * int(rng.integers(1, 12))
        pats.append(p); txts.append(bytes(t))
    b = gabgen.pairs_from_lists(pats, txts)
    same(eng.align(b), pyoracle.wfa(b))


def test_directory_in_lds_variant(eng, monkeypatch):
    """GAB_WFA_NO_STATIC=1: complete mode through the kernels that keep the directory in LDS (what adaptive mode always uses),
    with one-byte and with int16 offsets"""
    monkeypatch.setenv("GAB_WFA_NO_STATIC", "1")
    for seed, n, mode, plen in ((61, 30000, 0, 151), (62, 5000, 1, 300)):
        batch = gabgen.pairs(seed, n, mode, plen)
        want = pyoracle.wfa(batch, want_cells=True)
        same(eng.align(batch), want)
        assert eng.last_stats()["work"] == want[4]


def test_long_sequences_global_path(eng):
    """sequences beyond the LDS limit and scores beyond the LDS pools -> global-history kernel"""
    rng = np.random.default_rng(11)
    pats, txts = [], []
    for n, err in ((2500, 0.01), (6000, 0.03), (1500, 0.25), (300, 0.6)):
        p = rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
        t = bytearray()
        for c in p:
            r = rng.random()
            if r < err / 3: continue
            if r < 2 * err / 3: t.append(b"ACGT"[int(rng.integers(0, 4))])
            t.append(c if r > err else b"ACGT"[int(rng.integers(0, 4))])
        pats.append(p); txts.append(bytes(t))
    b = gabgen.pairs_from_lists(pats, txts)
    same(eng.align(b), pyoracle.wfa(b))
    assert eng.last_stats()["requeued"] >= 1


def _mutated(rng, n, err):
    p = rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
    t = bytearray()
    for c in p:
        r = rng.random()
        if r < err / 3: continue
        if r < 2 * err / 3: t.append(b"ACGT"[int(rng.integers(0, 4))])
        t.append(c if r > err else b"ACGT"[int(rng.integers(0, 4))])
    return p, bytes(t)


@pytest.mark.parametrize("tmax,pen,red", [(187, (4, 6, 2), None), (188, (4, 6, 2), None), (187, (1, 1, 1), None), (187, (2, 3, 1), None),
                                          (195, (4, 6, 2), (10, 50)), (196, (4, 6, 2), (10, 50))])
def test_one_byte_history_limits(tmax, pen, red):
    """the first tier keeps offsets as one byte (value + 10) when max text length + its score cap + 2 <= 245 (187 in
    complete mode, 195 in adaptive mode) and as int16 above: batches on both sides of the switch, with texts at the
    limit, scores on both sides of the tier's cap and offsets that run past the end of the text (+1 per score step on
    the diagonals beyond the last one), under penalty sets that visit every score"""
    from genarchbench_amd.wfa import AffineWavefronts
    rng = np.random.default_rng(tmax)
    pats, txts = [], []
    while len(pats) < 3000:
        n = int(rng.integers(tmax - 25, tmax + 1))
        p, t = _mutated(rng, n, float(rng.choice([0.0, 0.01, 0.03, 0.06, 0.12])))
        if len(t) > tmax: t = t[:tmax]
        pats.append(p); txts.append(t)
    pats += [b"ACGT" * 40, b"A" * (tmax - 30), b"ACGTTGCA" * 20]
    txts += [(b"ACGT" * 47)[:tmax], b"A" * tmax, (b"ACGTTGCA" * 24)[:tmax]]          # text at the limit, pure insertions
    b = gabgen.pairs_from_lists(pats, txts)
    assert int(b.txt_len.max()) == tmax
    kw = {} if red is None else dict(min_wavefront_length=red[0], max_distance_threshold=red[1])
    e = AffineWavefronts(*pen, **kw)
    same(e.align(b), pyoracle.wfa(b, pen) if red is None else pyoracle.wfa(b, pen, reduction=red))
    assert e.last_stats()["requeued"] >= 1          # some pairs exceed the first tier
    e.close()


@pytest.mark.parametrize("slots", [1001, 1002, 3, 0])
def test_resumed_and_restarted_pairs_in_one_wave(monkeypatch, slots):
    """the second static launch continues the pairs the first ran out of room for from the row they had reached -- for the
    first `slots` of them; the others start over at row 0.  A slot count that is not a multiple of the four pairs of a
    wave puts both kinds into ONE wave, whose row index is then no longer wave-uniform: the kernel runs
    such a wave's groups one after the other.  High-divergence batch: most pairs leave the first tier."""
    from genarchbench_amd.wfa import AffineWavefronts
    monkeypatch.setenv("GAB_WFA_SLOTS", str(slots))
    rng = np.random.default_rng(1000 + slots)
    pats, txts = [], []
    for _ in range(6000):
        p, t = _mutated(rng, int(rng.integers(120, 152)), float(rng.choice([0.01, 0.08, 0.1, 0.12])))
        pats.append(p); txts.append(t[:180])
    b = gabgen.pairs_from_lists(pats, txts)
    e = AffineWavefronts()
    same(e.align(b), pyoracle.wfa(b))
    assert e.last_stats()["requeued"] > max(slots, 1) + 8          # more pairs overflowed than there are slots
    e.close()


def _rle(ops):
    out, k, n = bytearray(), 0, len(ops)
    while k < n:
        r = k
        while r < n and ops[r] == ops[k]:
            r += 1
        out += b"%d%c" % (r - k, ops[k]); k = r
    return bytes(out)


@pytest.mark.parametrize("name", ["wfa_bench", "wfa_adv"])
def test_packed_output_golden(eng, name):
    """gab_wfa_run_packed returns the text edit_cigar_print writes: byte-identical to the reference's output lines"""
    batch = gabgen.read_pairs_text(f"{GOLDEN}/{name}.in.txt")
    want = [l.split(b" ", 1)[1] if b" " in l else b"" for l in open(f"{GOLDEN}/{name}.expected.txt", "rb").read().splitlines()]
    text, off, ln, sc = eng.align_packed(batch)
    got = [text[off[i]:off[i] + ln[i]].tobytes() for i in range(batch.n)]
    assert got == want
    assert int(ln.sum()) == len(text)                          # packed without gaps
    np.testing.assert_array_equal(sc, pyoracle.wfa(batch)[3])


def test_packed_output_vs_unpacked_and_capacity(eng):
    """same alignments as gab_wfa_run, run-length encoded; a buffer that is too small is reported with the size that fits"""
    from genarchbench_amd._lib import GabError
    rng = np.random.default_rng(77)
    pats, txts = [b"", b"A", b"", b"ACGT" * 300, b"A" * 1200], [b"", b"", b"ACG", b"ACGA" * 300, b"A" * 1000]     # empty CIGAR, pure I / D, runs >= 1000
    for _ in range(5000):
        p, t = _mutated(rng, int(rng.integers(1, 260)), float(rng.choice([0.0, 0.02, 0.1, 0.3])))
        pats.append(p); txts.append(t)
    b = gabgen.pairs_from_lists(pats, txts)
    ops, ooff, oln, osc = eng.align(b)
    text, off, ln, sc = eng.align_packed(b)
    np.testing.assert_array_equal(sc, osc)
    for i in range(b.n):
        assert text[off[i]:off[i] + ln[i]].tobytes() == _rle(ops[ooff[i]:ooff[i] + oln[i]].tobytes()), i
    assert ln[0] == 0 and ln[1] == 2 and ln[2] == 2                # "", "1D", "3I"
    # one long pair among short ones: the operation room on the device switches from a fixed stride to exact offsets
    lp, lt = _mutated(rng, 30000, 0.01)
    mixed = gabgen.pairs_from_lists(pats[:600] + [lp], txts[:600] + [lt])
    mo, moff, mln, msc = eng.align(mixed)
    mt, mtoff, mtln, mtsc = eng.align_packed(mixed)
    np.testing.assert_array_equal(mtsc, msc)
    for i in (0, 5, 599, 600):
        assert mt[mtoff[i]:mtoff[i] + mtln[i]].tobytes() == _rle(mo[moff[i]:moff[i] + mln[i]].tobytes()), i
    need = int(ln.sum())
    with pytest.raises(GabError) as e:
        eng.align_packed(b, capacity=need - 1)
    assert e.value.code == -34
    t2, off2, ln2, _ = eng.align_packed(b, capacity=need)      # exactly enough
    assert len(t2) == need and np.array_equal(ln2, ln)


def test_other_penalties():
    from genarchbench_amd.wfa import AffineWavefronts
    b = gabgen.pairs(67, 5000, 1, 120)
    for pen in [(1, 1, 1), (2, 3, 1), (5, 8, 3), (3, 1, 4)]:
        e = AffineWavefronts(*pen)
        same(e.align(b), pyoracle.wfa(b, pen))
        e.close()


# ---- adaptive reduction ----------------------------------------------------------------------
@pytest.mark.parametrize("red", [(10, 10), (5, 3), (1, 0)])
def test_adaptive_golden(red):
    from genarchbench_amd.wfa import AffineWavefronts
    batch = gabgen.read_pairs_text(f"{GOLDEN}/wfa_adv.in.txt")
    want = read_cigars(f"{GOLDEN}/wfa_adv.adaptive_{red[0]}_{red[1]}.expected.txt")
    e = AffineWavefronts(min_wavefront_length=red[0], max_distance_threshold=red[1])
    assert pyoracle.wfa_cigars(e.align(batch)) == want
    e.close()


@pytest.mark.parametrize("red,seed,n,mode,plen,pen", [((10, 50), 71, 40000, 0, 151, (4, 6, 2)), ((5, 3), 72, 20000, 1, 300, (4, 6, 2)),
                                                      ((3, 1), 73, 3000, 1, 900, (4, 6, 2)), ((0, -1), 74, 5000, 1, 120, (4, 6, 2)),
                                                      ((2, 5), 75, 5000, 1, 200, (2, 3, 1)), ((20, 2), 76, 5000, 1, 250, (5, 8, 3)),
                                                      ((1, 0), 77, 65, 1, 40, (3, 1, 4))])
def test_adaptive_vs_oracle(red, seed, n, mode, plen, pen):
    from genarchbench_amd.wfa import AffineWavefronts
    batch = gabgen.pairs(seed, n, mode, plen)
    want = pyoracle.wfa(batch, pen, want_cells=True, reduction=red)
    e = AffineWavefronts(*pen, min_wavefront_length=red[0], max_distance_threshold=red[1])
    same(e.align(batch), want)
    assert e.last_stats()["work"] == want[4]
    e.close()


def test_adaptive_long_sequences_global_path():
    """the int32 / global-history kernel in adaptive mode, and the edge inputs"""
    from genarchbench_amd.wfa import AffineWavefronts
    rng = np.random.default_rng(12)
    pats, txts = [b"A", b"ACGT", b"", b"XXYY", b"A" * 200], [b"A", b"", b"ACGT", b"YYXX", b"A" * 120]
    for n, err in ((2500, 0.02), (6000, 0.05), (1500, 0.25), (300, 0.6)):
        p = rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
        t = bytearray()
        for c in p:
            r = rng.random()
            if r < err / 3: continue
            if r < 2 * err / 3: t.append(b"ACGT"[int(rng.integers(0, 4))])
            t.append(c if r > err else b"ACGT"[int(rng.integers(0, 4))])
        pats.append(p); txts.append(bytes(t))
    b = gabgen.pairs_from_lists(pats, txts)
    for red in [(10, 50), (5, 3), (0, -1)]:
        e = AffineWavefronts(min_wavefront_length=red[0], max_distance_threshold=red[1])
        same(e.align(b), pyoracle.wfa(b, reduction=red))
        e.close()


def test_device_resident(eng):
    import torch
    from genarchbench_amd.wfa import ops_layout
    batch = gabgen.pairs(68, 30000, 0, 151)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    off, total = ops_layout(batch)
    ops = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
    ln = torch.zeros(batch.n, dtype=torch.int32, device=dev); sc = torch.zeros_like(ln)
    eng.run_device(t(batch.pat), t(batch.pat_off), t(batch.pat_len), t(batch.txt), t(batch.txt_off), t(batch.txt_len),
                   ops, t(off), ln, sc, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    same((ops.cpu().numpy(), off, ln.cpu().numpy(), sc.cpu().numpy()), pyoracle.wfa(batch))


def test_packed_output_from_device_resident_pairs(eng):
    """gab_wfa_run_packed_device (the drivers' GPU-parse path): pairs already on the device, the printed text back to host arrays
    -- the same text as gab_wfa_run_packed gives for host pointers, and the oracle's scores; too little room is GAB_ERANGE with the
    size that fits"""
    import torch
    from genarchbench_amd._lib import GabError
    from genarchbench_amd.wfa import ops_layout
    batch = gabgen.pairs(69, 20000, 0, 151)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    off, total = ops_layout(batch)
    ops = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
    args = (t(batch.pat), t(batch.pat_off), t(batch.pat_len), t(batch.txt), t(batch.txt_off), t(batch.txt_len), ops, t(off))
    text, toff, tln, sc, need = eng.run_packed_device(*args, capacity=total // 4 + 4096)
    wt, woff, wln, wsc = eng.align_packed(batch)
    np.testing.assert_array_equal(sc, wsc)
    np.testing.assert_array_equal(sc, pyoracle.wfa(batch)[3])
    np.testing.assert_array_equal(tln, wln)
    assert need == int(tln.sum()) == len(text)
    for i in range(batch.n):
        assert text[toff[i]:toff[i] + tln[i]].tobytes() == wt[woff[i]:woff[i] + wln[i]].tobytes(), i
    with pytest.raises(GabError) as e:
        eng.run_packed_device(*args, capacity=need - 1)
    assert e.value.code == -34
    t2, _, ln2, _, need2 = eng.run_packed_device(*args, capacity=need)       # exactly enough
    assert need2 == need and np.array_equal(ln2, tln)
