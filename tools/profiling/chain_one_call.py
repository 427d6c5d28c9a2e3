#!/usr/bin/env python3
"""Latency of chain / fast-chain calls through gab_chain_run_device (profiles/r03_chain_latency.md): batches of 1 .. 512 generated
calls of exactly N anchors, kernel time from the library's HIP events (best of three after a warm-up).
    python tools/profiling/chain_one_call.py [label] [mode 0|1] [anchors per call] [call counts ...]
GAB_CHAIN_FAST_MIN=1 puts every call into the latency form, GAB_CHAIN_FAST_CALLS=0 none; GAB_LIB_PATH selects a knock-out build
(-DGAB_KO_SEARCH / _FAR / _G / _MAIN in chain.hip: parts of chain_fast_kernel removed -- wrong results, only the time counts)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np      # noqa: E402
import torch            # noqa: E402
from tools import gabgen                               # noqa: E402
from genarchbench_amd.chain import ChainEngine         # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ""
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
counts = [int(v) for v in sys.argv[4:]] or [1, 2, 4, 8, 16, 32, 256]
dev = torch.device("cuda", 0)
for ncalls in counts:
    b = gabgen.chain(5, ncalls, 0, n, n)
    x = torch.from_numpy(b.x.view(np.int64)).to(dev); y = torch.from_numpy(b.y.view(np.int64)).to(dev)
    sc = torch.empty(b.nanchors, dtype=torch.int32, device=dev); pa = torch.empty_like(sc)
    e = ChainEngine(device=0)
    ms = []
    for _ in range(4):
        e.run_device(mode, x, y, b.call_off, b.hdr, sc, pa)
        ms.append(e.last_stats()["kernel_ms"])
    ev = e.last_stats()["evals"]
    k = min(ms[1:])
    print(f"{label}: {ncalls} call(s) of {n} anchors, {ev / b.nanchors:.0f} predecessor evaluations per anchor: {k:.3f} ms = {k * 1e3 / n:.3f} us per anchor of a call", flush=True)
    e.close()
