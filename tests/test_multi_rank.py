"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- id-range sharding with no data-path collective,
barrier + MAX(elapsed) + SUM(units) aggregation.  The GPU call is stood in for by the oracle here."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """a port nobody listens on right now (a fixed one collides with a lingering or parallel run: EADDRINUSE)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])

WORKER = textwrap.dedent("""
    import os, sys, json, time
    sys.path.insert(0, %r)
    import numpy as np
    import torch, torch.distributed as dist
    from genarchbench_amd.shard import rank_world, shard_range, aggregate
    from tools import gabgen
    from oracle import pyoracle
    rank, local, world = rank_world()
    dist.init_process_group("gloo")
    items = 3000
    first, n = shard_range(rank, world, items)
    batch = gabgen.bsw(123, n, 1, first=first)                 # this rank's shard only
    dist.barrier()
    t0 = time.perf_counter()
    scores = pyoracle.bsw(batch, threads=1)[:, 0]              # stands in for the per-GPU engine
    time.sleep(0.05 * (rank + 1))                              # ranks finish at different times
    dist.barrier()
    el = time.perf_counter() - t0
    el_all, units = aggregate(el, n, dist)
    out = [None] * world
    dist.all_gather_object(out, (rank, first, scores.tolist(), el))
    if rank == 0:
        json.dump({"elapsed": el_all, "units": units, "parts": out}, open(sys.argv[1], "w"))
    dist.destroy_process_group()
""") % ROOT


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"; script.write_text(WORKER)
    out = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", free_port(), str(script), str(out)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    res = json.load(open(out))
    assert res["units"] == 6000
    assert res["elapsed"] >= max(p[3] for p in res["parts"]) - 1e-6     # MAX over ranks
    # the two shards together are exactly the single-process result on ids 0..5999
    sys.path.insert(0, ROOT)
    from tools import gabgen
    from oracle import pyoracle
    whole = pyoracle.bsw(gabgen.bsw(123, 6000, 1))[:, 0]
    got = np.zeros(6000, np.int32)
    for rank, first, sc, _ in res["parts"]:
        got[first:first + len(sc)] = sc
    np.testing.assert_array_equal(got, whole)


def test_shard_ranges_tile_the_id_space():
    from genarchbench_amd.shard import shard_range, shard_strong
    for world in (1, 2, 4, 8):
        spans = [shard_range(r, world, 1000) for r in range(world)]
        assert [s[0] for s in spans] == [1000 * r for r in range(world)] and all(s[1] == 1000 for s in spans)
    # strong: one fixed range, split; the shares tile it exactly and differ by at most one item
    for total in (0, 1, 7, 1000, 10_000_001):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_strong(r, world, total) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            assert all(spans[r][0] + spans[r][1] == spans[r + 1][0] for r in range(world - 1))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_deal_longest_first_balances_and_tiles():
    from genarchbench_amd.shard import deal_longest_first
    sys.path.insert(0, ROOT)
    from tools import gabgen
    sizes = gabgen.chain_sizes(5, 10000, 0, 50, 60000)           # the anchor counts of chain-large
    for world in (1, 2, 4, 8):
        deal = deal_longest_first(sizes, world)
        allids = np.concatenate(deal)
        assert len(allids) == len(sizes) and np.array_equal(np.sort(allids), np.arange(len(sizes)))      # every call exactly once
        loads = np.array([sizes[d].sum() for d in deal])
        assert loads.max() - loads.min() <= sizes.max()          # LPT: no rank is more than one call ahead
        assert loads.max() <= 1.01 * sizes.sum() / world
        longest = np.argsort(-sizes, kind="stable")[:world]      # the `world` longest calls land on `world` different ranks
        assert len({next(r for r, d in enumerate(deal) if i in d) for i in longest}) == world
    assert [list(d) for d in deal_longest_first([5, 5, 5, 5], 2)] == [[0, 2], [1, 3]]      # ties: lower id, lower rank first


STRONG_WORKER = textwrap.dedent("""
    import os, sys, json, time
    sys.path.insert(0, %r)
    import numpy as np
    import torch, torch.distributed as dist
    from genarchbench_amd.shard import rank_world, shard_strong, deal_longest_first, aggregate
    from tools import gabgen
    from oracle import pyoracle
    rank, local, world = rank_world()
    dist.init_process_group("gloo")
    T, C = 5001, 60                                            # ONE fixed input: T bsw pairs, C chain calls
    first, n = shard_strong(rank, world, T)
    batch = gabgen.bsw(321, n, 1, first=first)
    ids = deal_longest_first(gabgen.chain_sizes(77, C, 0, 50, 3000), world)[rank]
    calls = gabgen.chain_ids(77, ids, 0, 50, 3000)
    dist.barrier()
    t0 = time.perf_counter()
    scores = pyoracle.bsw(batch, threads=1)[:, 0]              # stands in for the per-GPU engines
    cs, cp = pyoracle.chain(calls, 0, threads=1)
    dist.barrier()
    el = time.perf_counter() - t0
    el_all, units = aggregate(el, n, dist)
    _, seeds = aggregate(el, calls.nanchors, dist)
    out = [None] * world
    dist.all_gather_object(out, (rank, first, scores.tolist(), ids.tolist(), calls.hdr["n"].tolist(), cs.tolist(), cp.tolist()))
    if rank == 0:
        json.dump({"units": units, "seeds": seeds, "parts": out}, open(sys.argv[1], "w"))
    dist.destroy_process_group()
""") % ROOT


def test_two_ranks_gloo_strong(tmp_path):
    """strong scaling: the shares of the ranks are the fixed input, nothing more, nothing twice; results checked against the
    oracle on the whole input (bsw: contiguous id ranges; chain: calls dealt longest first)"""
    script = tmp_path / "worker.py"; script.write_text(STRONG_WORKER)
    out = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", free_port(), str(script), str(out)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    res = json.load(open(out))
    sys.path.insert(0, ROOT)
    from tools import gabgen
    from oracle import pyoracle
    T, C = 5001, 60
    assert res["units"] == T
    whole = pyoracle.bsw(gabgen.bsw(321, T, 1))[:, 0]
    got = np.full(T, -99, np.int32)
    for rank, first, sc, *_ in res["parts"]:
        assert (got[first:first + len(sc)] == -99).all()
        got[first:first + len(sc)] = sc
    np.testing.assert_array_equal(got, whole)
    full = gabgen.chain(77, C, 0, 50, 3000)
    ws, wp = pyoracle.chain(full, 0)
    assert res["seeds"] == full.nanchors
    seen = set()
    for rank, _, _, ids, ns, cs, cp in res["parts"]:
        o = 0
        for i, n in zip(ids, ns):
            assert i not in seen; seen.add(i)
            a = int(full.call_off[i])
            assert n == full.hdr["n"][i]
            np.testing.assert_array_equal(cs[o:o + n], ws[a:a + n]); np.testing.assert_array_equal(cp[o:o + n], wp[a:a + n])
            o += n
    assert seen == set(range(C))


@pytest.mark.gpu
def test_bench_gpus_flag_is_binding():
    """`bench.py --gpus N` either runs N ranks or fails: it never reports a 1-GPU number as an N-GPU one.
    No launcher around it: it starts the ranks itself (as a child, before touching a GPU) and needs N visible GPUs;
    under a launcher WORLD_SIZE must equal --gpus."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) are visible" in r.stderr + r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr + r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["strong", "weak"])
@pytest.mark.parametrize("workload,items", [("bsw", 200_000), ("chain", 600)])
def test_two_ranks_real_engines_one_gpu(workload, items, scaling):
    """the N > 1 path of bench.py with REAL engine handles: two ranks (torch.distributed.run, as the driver launches it),
    each with its own handle on its own id range and checking its own shard against the oracle; on a one-GPU box both use
    GPU 0 and rendezvous over gloo (GAB_BENCH_SHARE_GPU=1 -- the line says it is not a 2-GPU measurement)"""
    import json
    env = dict(os.environ, GAB_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--items", str(items),
                        "--scaling", scaling, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-roi"],
                       env=dict(env, GAB_BENCH_SUITE_JSON=os.devnull), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = r.stdout.splitlines()[-1]                   # the contract: the headline is the LAST line of stdout
    assert len(line) < 4096
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and "TEST MODE" in d["config"]["sharding"]
    assert d["parity"].startswith("bit-exact")
    # strong: the two ranks together processed the fixed input once; weak: one input each
    assert d["config"]["total_items"] == (items if scaling == "strong" else 2 * items)
    # whole-job value = units of BOTH ranks / max-over-ranks time
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--items", str(items), "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-host-roi"], env=dict(os.environ, GAB_BENCH_SUITE_JSON=os.devnull),
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.splitlines()[-1])
    assert d1["n_gpus"] == 1 and d1["scaling"] == "weak"
    # value x ms_per_step = units per step of the whole job: once the 1-GPU input (strong) or twice (weak); chain's units
    # are seeds, and the two halves of a weak run are different calls
    units, units1 = d["value"] * d["ms_per_step"], d1["value"] * d1["ms_per_step"]
    if scaling == "strong":
        assert units == pytest.approx(units1, rel=0.02)
    else:
        assert units == pytest.approx(2 * units1, rel=0.02 if workload == "bsw" else 0.5)
