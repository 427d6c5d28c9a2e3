"""GPU parity: chain / fast-chain HIP kernels (through the C ABI) vs the oracle and the golden vectors."""
import numpy as np
import pytest

from oracle import pyoracle
from tools import gabgen
from tests.util import GOLDEN, read_chain_output

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default-split", "latency-form-for-all", "throughput-form-for-all", "table-form-for-all"])
def kernel_choice(request, monkeypatch):
    """chain: the longest calls of a batch (>= 4096 anchors) run in the latency form (chain_fast_kernel: geometry prepared by the
    helper waves, key-max fold, certificate by popcount), the rest in the throughput form.  Every test of this file runs with
    the default split (by the batch-shape rule of gab_chain_run_device), with EVERY call in the latency form and with none -- chain and fast-chain alike.
    r04: and with every call in the TABLE form (chain_tab.hip: geometry bytes by any CU, the call's workgroup only folds); calls it
    cannot take (a byte does not hold their gap costs, x not ascending, a missed max_skip certificate) come back through the latency form."""
    if request.param == "latency-form-for-all":
        monkeypatch.setenv("GAB_CHAIN_FAST_MIN", "1"); monkeypatch.setenv("GAB_CHAIN_FAST_CALLS", "1000000000"); monkeypatch.setenv("GAB_CHAIN_TAB", "0")
    elif request.param == "throughput-form-for-all":
        monkeypatch.setenv("GAB_CHAIN_FAST_CALLS", "0"); monkeypatch.setenv("GAB_CHAIN_TAB", "0")
    elif request.param == "table-form-for-all":
        monkeypatch.setenv("GAB_CHAIN_TAB_MIN", "1")
    return request.param


@pytest.fixture(scope="module")
def eng():
    from genarchbench_amd.chain import ChainEngine
    e = ChainEngine()
    yield e
    e.close()


@pytest.mark.parametrize("name", ["chain_bench", "chain_dense"])
@pytest.mark.parametrize("mode,tag", [(0, "chain"), (1, "fastchain")])
def test_golden(eng, name, mode, tag):
    batch = gabgen.read_chain_text(f"{GOLDEN}/{name}.in.txt")
    ws, wp = read_chain_output(f"{GOLDEN}/{name}.{tag}.expected.txt")
    s, p = eng.host_chain_kernel(batch, mode)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("seed,ncalls,gmode,nmin,nmax", [(31, 200, 0, 50, 20000), (32, 30, 1, 500, 9000),
                                                         (33, 3, 1, 40000, 70000), (34, 64, 0, 1, 70)])
def test_vs_oracle(eng, mode, seed, ncalls, gmode, nmin, nmax):
    batch = gabgen.chain(seed, ncalls, gmode, nmin, nmax)
    ws, wp, ev = pyoracle.chain(batch, mode, want_evals=True)
    s, p = eng.host_chain_kernel(batch, mode)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)
    # predecessor evaluations performed: the reference's visits are a lower bound (chain resolves most anchors by the plain
    # maximum over the whole window, which the max_skip scan would have left early), the window sizes an upper bound
    ev_gpu = eng.last_stats()["evals"]
    assert ev_gpu == ev if mode == 1 else ev <= ev_gpu <= 2 * sum(min(i, 5000) for n in batch.hdr["n"] for i in range(int(n)))


def test_table_form_patches_and_hands_back(eng, monkeypatch, capfd, kernel_choice):
    """chain, table form: an anchor whose max_skip certificate misses is scanned again the reference's way (chain_exact_global); when
    that really gives another result the anchor is PATCHED and the call starts again from its block (seed 31 holds such a call:
    anchor 1881 of a 3306-anchor call takes a predecessor 49 back instead of the plain maximum's); a call with more than eight such
    anchors (the dense golden set, where chain and fast-chain differ in hundreds of lines) goes back to the kernels of chain.hip"""
    if kernel_choice != "table-form-for-all":
        pytest.skip("table form only")
    monkeypatch.setenv("GAB_CHAIN_TRACE", "1")
    batch = gabgen.chain(31, 200, 0, 50, 20000)
    ws, wp = pyoracle.chain(batch, 0)
    capfd.readouterr()
    s, p = eng.host_chain_kernel(batch, 0)
    err = capfd.readouterr().err
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)
    line = [l for l in err.splitlines() if "eligible" in l][-1]
    assert " 0 handed back" in line and "exact re-scans" in line and int(line.split(";")[1].split()[0]) >= 1, line
    dense = gabgen.read_chain_text(f"{GOLDEN}/chain_dense.in.txt")
    ws, wp = read_chain_output(f"{GOLDEN}/chain_dense.chain.expected.txt")
    s, p = eng.host_chain_kernel(dense, 0)
    err = capfd.readouterr().err
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)
    line = [l for l in err.splitlines() if "eligible" in l][-1]
    assert int(line.split("eligible,")[1].split()[0]) >= 1, line          # handed back: too many patches


@pytest.mark.parametrize("mode", [0, 1])
def test_table_too_small_for_the_batch(mode, monkeypatch, capfd, kernel_choice):
    """GAB_CHAIN_TAB_MB=2: the table holds 2 048 groups of 16 rows -- a few short calls; the calls that find no room (the list is
    sorted longest first: the long ones take what there is) keep the kernels of chain.hip, and the batch is the oracle's"""
    if kernel_choice != "table-form-for-all":
        pytest.skip("table form only")
    from genarchbench_amd.chain import ChainEngine
    monkeypatch.setenv("GAB_CHAIN_TAB_MB", "2")
    monkeypatch.setenv("GAB_CHAIN_TRACE", "1")
    e = ChainEngine()                                           # (the budget is fixed at a handle's first table-form call)
    batch = gabgen.chain(35, 120, 0, 50, 6000)
    ws, wp = pyoracle.chain(batch, mode)
    capfd.readouterr()
    s, p = e.host_chain_kernel(batch, mode)
    err = capfd.readouterr().err
    e.close()
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)
    line = [l for l in err.splitlines() if "eligible" in l][-1]
    assert int(line.split("eligible,")[1].split()[0]) >= 1, line          # handed back: no room


def test_walk_kernel_variant(eng, monkeypatch):
    """GAB_CHAIN_KERNEL=walk: the per-anchor walk kernel kept for A/B runs gives the reference's result too"""
    monkeypatch.setenv("GAB_CHAIN_KERNEL", "walk")
    batch = gabgen.chain(39, 60, 1, 500, 9000)
    ws, wp = pyoracle.chain(batch, 0)
    s, p = eng.host_chain_kernel(batch, 0)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)


@pytest.mark.parametrize("helpers", [3, 5, 7])
@pytest.mark.parametrize("mode", [0, 1])
def test_every_helper_count(eng, mode, helpers, monkeypatch):
    """the library picks three helper waves per call for throughput-bound batches and seven for batches bound by their
    longest call; every instantiation has to give the reference's result (the environment variable pins the choice)"""
    monkeypatch.setenv("GAB_CHAIN_HELPERS", str(helpers))
    for batch in (gabgen.chain(35, 40, 0, 50, 30000), gabgen.chain(36, 25, 1, 500, 9000)):
        ws, wp = pyoracle.chain(batch, mode)
        s, p = eng.host_chain_kernel(batch, mode)
        np.testing.assert_array_equal(s, ws)
        np.testing.assert_array_equal(p, wp)


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("mode", [0, 1])
def test_big_batch_host_paths(eng, mode, pinned, monkeypatch):
    """gab_chain_run on big batches: page-locked arrays take the fed path (a kernel fetches the anchors longest call first,
    the DP workgroups wait per call and write their results through to the host arrays), pageable ones the copy-engine
    path with three streams; the threshold is lowered so that both run here, on calls that miss the block kernel's
    per-call shortcuts as well (several segment ids, x spread over more than 2^31, wide bands)"""
    monkeypatch.setenv("GAB_CHAIN_FEED_MIN", "1000")
    rng = np.random.default_rng(23)
    Y = lambda q, span=15, seg=0: (np.uint64(seg) << np.uint64(48)) | (np.uint64(span) << np.uint64(32)) | np.uint64(q)
    x = np.sort(np.concatenate([rng.integers(0, 20000, 1500), (1 << 33) + rng.integers(0, 20000, 1500)])).astype(np.uint64)
    q = (x.astype(np.int64) % 20000 + rng.integers(-30, 30, 3000)).clip(0)
    odd = gabgen.chain_from_calls([(15.0, 5000, 5000, 500, 3, x, np.array([Y(int(v), 15, int(g)) for v, g in zip(q, rng.integers(0, 3, 3000))], np.uint64)),
                                   (15.0, 5000, 5000, 3000, 1, x % np.uint64(50000), np.array([Y(int(v)) for v in q], np.uint64))])
    for batch in (gabgen.chain(37, 300, 0, 1, 20000), gabgen.chain(38, 40, 1, 500, 9000), odd):
        ws, wp = pyoracle.chain(batch, mode)
        s, p = eng.host_chain_kernel(batch, mode, pinned=pinned)
        np.testing.assert_array_equal(s, ws)
        np.testing.assert_array_equal(p, wp)


@pytest.mark.parametrize("mode", [0, 1])
def test_fed_path_gives_up_and_the_copy_engines_take_over(eng, mode, monkeypatch, capfd, kernel_choice):
    """the branch nobody runs: a workgroup of the fed DP kernel whose anchors never arrive -- here: the gather kernel
    is told not to publish the longest call's word (GAB_CHAIN_FEED_GIVEUP) and the wait is shortened to ~20 ms -- gives up, sets the
    abort word, every other wait follows, the grid drains, and gab_chain_run runs the batch again through the copy engines
    (chain_run_overlapped): the reference's result, and the note once"""
    if kernel_choice != "default-split":
        pytest.skip("one kernel choice is enough for the host path")
    monkeypatch.setenv("GAB_CHAIN_FEED_MIN", "1000")
    monkeypatch.setenv("GAB_CHAIN_FEED_GIVEUP", "1")
    batch = gabgen.chain(37, 300, 0, 1, 20000)
    ws, wp = pyoracle.chain(batch, mode)
    capfd.readouterr()
    s, p = eng.host_chain_kernel(batch, mode, pinned=True)
    err = capfd.readouterr().err
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)
    assert err.count("gave up waiting for its anchors") == 1, err[-500:]


@pytest.mark.parametrize("mode", [0, 1])
def test_fed_path_two_handles_on_one_gpu(mode, monkeypatch, kernel_choice):
    """two handles of one GPU in the fed path AT ONCE (a driver's queue workers run side by side): each
    has its own gather stream on its own CU per XCD, its own residency probe through host memory and its own facts words -- three
    rounds of two concurrent calls on different batches, every result the oracle's"""
    if kernel_choice != "default-split":
        pytest.skip("one kernel choice is enough for the host path")
    import threading
    from genarchbench_amd.chain import ChainEngine
    monkeypatch.setenv("GAB_CHAIN_FEED_MIN", "1000")
    batches = [gabgen.chain(41, 400, 0, 1, 20000), gabgen.chain(42, 60, 1, 500, 12000)]
    want = [pyoracle.chain(b, mode) for b in batches]
    engines = [ChainEngine(), ChainEngine()]
    errors = []

    import dataclasses
    # (every worker its own copies of the arrays, as a driver's workers have their own slabs: the binding page-locks the arrays of
    # a call and unlocks them after it -- shared arrays would be unlocked under the other worker's running fetch kernel)
    own = [[dataclasses.replace(b, x=b.x.copy(), y=b.y.copy()) for b in batches] for _ in range(2)]

    def work(k):
        try:
            for rnd in range(3):
                b = own[k][(k + rnd) % 2]
                s, p = engines[k].host_chain_kernel(b, mode, pinned=True)
                ws, wp = want[(k + rnd) % 2]
                if not (np.array_equal(s, ws) and np.array_equal(p, wp)):
                    errors.append((k, rnd))
        except Exception as e:      # noqa: BLE001  (reported below, from the main thread)
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in engines:
        e.close()
    assert not errors, errors


@pytest.mark.parametrize("mode", [0, 1])
def test_gap_cost_table_limits(eng, mode):
    """the block kernels read the gap cost of a pair from a per-call table of bw + 2 entries when 0 <= bw <= 2046 and
    compute it otherwise: band widths on both sides of the limit, 0 and 1, with diagonal differences from 0 to beyond bw,
    fractional and large avg_qspan"""
    rng = np.random.default_rng(17)
    Y = lambda q, span=15, seg=0: (np.uint64(seg) << np.uint64(48)) | (np.uint64(span) << np.uint64(32)) | np.uint64(q)
    calls = []
    for bw, aq in ((0, 15.0), (1, 15.0), (500, 14.37), (2046, 15.0), (2047, 15.0), (2046, 250.5), (3000, 7.25)):
        x = np.sort(rng.integers(0, 60000, 2500)).astype(np.uint64)
        q = (x.astype(np.int64) // 2 + rng.integers(0, max(2 * bw, 4) + 50, 2500)).clip(0)      # dd = |dr - dq| up to ~2 bw
        calls.append((aq, 5000, 5000, bw, 1, x, np.array([Y(int(v), int(s)) for v, s in zip(q, rng.integers(1, 60, 2500))], np.uint64)))
    batch = gabgen.chain_from_calls(calls)
    ws, wp = pyoracle.chain(batch, mode)
    s, p = eng.host_chain_kernel(batch, mode)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)
    assert (wp >= 0).sum() > 1000          # the filter lets many pairs through: the gap cost matters


@pytest.mark.parametrize("mode", [0, 1])
def test_edge_calls(eng, mode):
    """empty call, single anchor, duplicates, unsorted x, window clamp (max_iter), huge max_dist"""
    rng = np.random.default_rng(5)
    Y = lambda q, span=15, seg=0: (np.uint64(seg) << np.uint64(48)) | (np.uint64(span) << np.uint64(32)) | np.uint64(q)
    calls = []
    calls.append((15.0, 5000, 5000, 500, 1, np.array([], np.uint64), np.array([], np.uint64)))
    calls.append((15.0, 5000, 5000, 500, 1, np.array([7], np.uint64), np.array([Y(3)], np.uint64)))
    x = np.sort(rng.integers(0, 50, 300)).astype(np.uint64)
    calls.append((15.0, 5000, 5000, 500, 1, x, np.array([Y(int(q)) for q in rng.integers(0, 60, 300)], np.uint64)))
    x = rng.integers(0, 4000, 500).astype(np.uint64)  # NOT sorted
    calls.append((14.5, 5000, 5000, 500, 1, x, np.array([Y(int(q), int(s)) for q, s in zip(rng.integers(0, 4000, 500), rng.integers(1, 40, 500))], np.uint64)))
    # 12000 anchors inside one max_dist_x window: the 5000-predecessor clamp applies
    x = np.sort(rng.integers(0, 3000, 12000)).astype(np.uint64)
    q = (x.astype(np.int64) + rng.integers(-20, 20, 12000)).clip(0)
    calls.append((15.0, 1000000, 1000000, 100000, 1, x, np.array([Y(int(v)) for v in q], np.uint64)))
    batch = gabgen.chain_from_calls(calls)
    ws, wp = pyoracle.chain(batch, mode)
    s, p = eng.host_chain_kernel(batch, mode)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)


@pytest.mark.parametrize("mode", [0, 1])
def test_generic_arithmetic_and_certificate_misses(eng, mode):
    """calls that miss the block kernel's per-call shortcuts -- x spread over more than 2^31 (64-bit differences), several
    segment ids (cross-segment gap rule), negative avg_qspan (gap cost rounding of negative values) -- and dense calls in
    which the plain maximum is NOT the reference's result for many anchors (the max_skip exit comes first): the exact
    scan has to take over"""
    rng = np.random.default_rng(9)
    Y = lambda q, span=15, seg=0: (np.uint64(seg) << np.uint64(48)) | (np.uint64(span) << np.uint64(32)) | np.uint64(q)
    calls = []
    # two clusters 2^33 apart inside one call, three segment ids
    x = np.sort(np.concatenate([rng.integers(0, 20000, 1500), (1 << 33) + rng.integers(0, 20000, 1500)])).astype(np.uint64)
    q = (x.astype(np.int64) % 20000 + rng.integers(-30, 30, 3000)).clip(0)
    calls.append((15.0, 5000, 5000, 500, 3, x, np.array([Y(int(v), 15, int(g)) for v, g in zip(q, rng.integers(0, 3, 3000))], np.uint64)))
    # same geometry, one segment, negative avg_qspan
    x = np.sort(rng.integers(0, 30000, 4000)).astype(np.uint64)
    q = (x.astype(np.int64) + rng.integers(-25, 25, 4000)).clip(0)
    calls.append((-7.5, 5000, 5000, 500, 1, x, np.array([Y(int(v)) for v in q], np.uint64)))
    batch = gabgen.chain_from_calls(calls)
    ws, wp = pyoracle.chain(batch, mode)
    s, p = eng.host_chain_kernel(batch, mode)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)
    # the generator's dense mode: 3.5 % of these anchors miss the certificate and 0.3 % end with a result different from the
    # plain maximum (measured with an instrumented copy of the oracle), on two- and one-segment calls alike
    dense = gabgen.chain(7, 60, 1, 1500, 6000)
    ws, wp = pyoracle.chain(dense, mode)
    s, p = eng.host_chain_kernel(dense, mode)
    np.testing.assert_array_equal(s, ws)
    np.testing.assert_array_equal(p, wp)


def test_device_resident(eng):
    import torch
    batch = gabgen.chain(41, 50, 0, 50, 5000)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(batch.x.view(np.int64)).to(dev); y = torch.from_numpy(batch.y.view(np.int64)).to(dev)
    sc = torch.zeros(batch.nanchors, dtype=torch.int32, device=dev); pa = torch.zeros_like(sc)
    for mode in (0, 1):
        eng.run_device(mode, x, y, batch.call_off, batch.hdr, sc, pa, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ws, wp = pyoracle.chain(batch, mode)
        np.testing.assert_array_equal(sc.cpu().numpy(), ws)
        np.testing.assert_array_equal(pa.cpu().numpy(), wp)


@pytest.mark.parametrize("pinned", [True, False])
def test_device_resident_results_written_through_to_the_host(eng, pinned):
    """gab_chain_run_device_through (the drivers' GPU-parse path): device arrays filled as always, and the same results in host arrays
    when the call returns -- written through by the DP kernel itself when the batch runs in the throughput form and the host arrays
    are page-locked (many equal calls: no call is waited for), copied at the end otherwise (pageable arrays; the other kernel forms of
    the dispatch modes this file runs under)"""
    import torch
    dev = torch.device("cuda:0")
    # (2 500 short calls: the throughput form writes through; 600 calls of 4 096 .. 6 000 anchors: all of them in the TABLE form, whose
    # fold writes through; 40 mixed calls: the latency form takes part, which does not -- copied at the end)
    for batch in (gabgen.chain(43, 2500, 0, 600, 1500), gabgen.chain(45, 600, 0, 4096, 6000), gabgen.chain(44, 40, 0, 50, 9000)):
        x = torch.from_numpy(batch.x.view(np.int64)).to(dev); y = torch.from_numpy(batch.y.view(np.int64)).to(dev)
        sc = torch.zeros(batch.nanchors, dtype=torch.int32, device=dev); pa = torch.zeros_like(sc)
        for mode in (0, 1):
            hs, hp = eng.run_device_through(mode, x, y, batch.call_off, batch.hdr, sc, pa, pinned=pinned, stream=torch.cuda.current_stream().cuda_stream)
            ws, wp = pyoracle.chain(batch, mode)
            np.testing.assert_array_equal(hs, ws)
            np.testing.assert_array_equal(hp, wp)
            np.testing.assert_array_equal(sc.cpu().numpy(), ws)
            np.testing.assert_array_equal(pa.cpu().numpy(), wp)
