import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tools import gabgen
from genarchbench_amd.chain import ChainEngine
cb = gabgen.chain(5, 10000, 0)
order = np.argsort(-cb.hdr["n"], kind="stable")[:64]
ce = ChainEngine(device=0)
res = []
for c in order:
    idx = np.array([c])
    sub = gabgen.ChainBatch(cb.hdr[idx].copy(), cb.call_off[idx].copy(), cb.x, cb.y)
    ce.host_chain_kernel(sub, 0); ce.host_chain_kernel(sub, 0)
    st = ce.last_stats(); n = int(sub.hdr["n"][0])
    res.append((st["kernel_ms"], n, st["evals"] / n))
res.sort(reverse=True)
for r in res[:8]: print("ms=%.1f n=%d evals/anchor=%.0f us/anchor=%.2f" % (r[0], r[1], r[2], r[0]*1e3/r[1]))
print("...", "ms=%.1f n=%d evals/anchor=%.0f" % res[-1])
