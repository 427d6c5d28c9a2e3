"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- id-range sharding with no data-path collective,
barrier + MAX(elapsed) + SUM(units) aggregation.  The GPU call is stood in for by the oracle here."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), str(out)],
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
    from genarchbench_amd.shard import shard_range
    for world in (1, 2, 4, 8):
        spans = [shard_range(r, world, 1000) for r in range(world)]
        assert [s[0] for s in spans] == [1000 * r for r in range(world)] and all(s[1] == 1000 for s in spans)


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


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("workload,items", [("bsw", 200_000), ("chain", 600)])
def test_two_ranks_real_engines_one_gpu(workload, items):
    """the N > 1 path of bench.py with REAL engine handles: two ranks (torch.distributed.run, as the driver launches it),
    each with its own handle on its own id range and checking its own shard against the oracle; on a one-GPU box both use
    GPU 0 and rendezvous over gloo (GAB_BENCH_SHARE_GPU=1 -- the line says it is not a 2-GPU measurement)"""
    import json
    env = dict(os.environ, GAB_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--items", str(items),
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-roi"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "TEST MODE" in d["config"]["sharding"]
    assert d["parity"].startswith("bit-exact")
    # whole-job value = units of BOTH ranks / max-over-ranks time
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--items", str(items), "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-host-roi"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["n_gpus"] == 1
    assert d["value"] * d["ms_per_step"] == pytest.approx(2 * d1["value"] * d1["ms_per_step"], rel=0.02 if workload == "bsw" else 0.5)
