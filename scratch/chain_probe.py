import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from tools import gabgen
from genarchbench_amd.chain import ChainEngine
dev = torch.device("cuda:0")
eng = ChainEngine()
def run(batch, mode, reps=3):
    x = torch.from_numpy(batch.x.view(np.int64)).to(dev); y = torch.from_numpy(batch.y.view(np.int64)).to(dev)
    sc = torch.zeros(batch.nanchors, dtype=torch.int32, device=dev); pa = torch.zeros_like(sc)
    ms = []
    for _ in range(reps):
        eng.run_device(mode, x, y, batch.call_off, batch.hdr, sc, pa, stream=torch.cuda.current_stream().cuda_stream)
        st = eng.last_stats(); ms.append(st["kernel_ms"])
    return min(ms), st["evals"]
for ncalls, nmin, nmax in [(1, 60000, 60000), (8, 60000, 60000), (256, 60000, 60000), (2048, 20000, 20000), (20000, 2000, 2000)]:
    b = gabgen.chain(5, ncalls, 0, nmin, nmax)
    for mode in (0, 1):
        ms, ev = run(b, mode)
        print(f"calls {ncalls:6d} n {nmax:6d} mode {mode}: {ms:9.3f} ms  {ms*1e3/nmax:8.3f} us/anchor-step  evals/anchor {ev/b.nanchors:6.1f}  {b.nanchors/ms/1e3:8.1f} Mseeds/s", flush=True)
