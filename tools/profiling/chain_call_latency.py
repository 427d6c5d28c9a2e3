import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tools import gabgen
from genarchbench_amd.chain import ChainEngine
cb = gabgen.chain(5, 10000, 0)
order = np.argsort(-cb.hdr["n"], kind="stable")
n = cb.hdr["n"][order]
print("longest calls:", n[:5], "total anchors", cb.nanchors, flush=True)
ce = ChainEngine(device=0)
MODE = int(os.environ.get('MODE', '0'))
def run(idx, label):
    sub = gabgen.ChainBatch(cb.hdr[idx].copy(), cb.call_off[idx].copy(), cb.x, cb.y)
    ce.host_chain_kernel(sub, MODE)
    ce.host_chain_kernel(sub, MODE)
    st = ce.last_stats()
    na = int(sub.hdr["n"].sum())
    print(f"{label}: calls={len(idx)} anchors={na} kernel_ms={st['kernel_ms']:.2f} evals/anchor={st['evals']/na:.1f} Mseeds/s={na/st['kernel_ms']/1e3:.1f}", flush=True)
run(order[:1], "top1")
run(order[:64], "top64")
run(order[:256], "top256")
run(order[:1024], "top1024")
run(order[:2048], "top2048")
run(order[2048:], "rest after 2048")
run(order[1024:], "rest after 1024")
run(order, "all")
