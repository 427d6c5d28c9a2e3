#!/usr/bin/env python3
"""Randomised parity of the chain / fast-chain kernel forms against the oracle (run on the GPU box, repo root): many seeds, both
generator modes, call-length ranges from a few anchors to 70 000, every dispatch mode (the default split, everything in the table
form, everything in the latency form, nothing in either) and both entry points of the device path (plain and written through).
Not part of pytest (minutes of oracle time); profiles/rNN_full_size_parity.md quotes the last run.

    python tests/stress_chain_table.py [first_seed [n_seeds]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GAB_TUNING_LIVE"] = "1"
from oracle import pyoracle      # noqa: E402  (the checker; this script is test infrastructure)
from tools import gabgen         # noqa: E402

MODES = {"default-split": {}, "table-form-for-all": {"GAB_CHAIN_TAB_MIN": "1"},
         "latency-form-for-all": {"GAB_CHAIN_FAST_MIN": "1", "GAB_CHAIN_FAST_CALLS": "1000000000", "GAB_CHAIN_TAB": "0"},
         "throughput-form-for-all": {"GAB_CHAIN_FAST_CALLS": "0", "GAB_CHAIN_TAB": "0"}}
KNOBS = sorted({k for v in MODES.values() for k in v})


def main():
    import torch
    from genarchbench_amd.chain import ChainEngine
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    dev = torch.device("cuda:0")
    eng = ChainEngine()
    rng = np.random.default_rng(first)
    bad = 0
    t0 = time.time()
    for k in range(count):
        seed = first + k
        gmode = int(rng.integers(0, 2))
        shape = int(rng.integers(0, 4))
        ncalls, nmin, nmax = [(400, 1, 3000), (60, 500, 20000), (6, 30000, 70000), (1500, 1, 300)][shape]
        b = gabgen.chain(seed, ncalls, gmode, nmin, nmax)
        x = torch.from_numpy(b.x.view(np.int64)).to(dev); y = torch.from_numpy(b.y.view(np.int64)).to(dev)
        sc = torch.zeros(b.nanchors, dtype=torch.int32, device=dev); pa = torch.zeros_like(sc)
        for mode in (0, 1):
            ws, wp = pyoracle.chain(b, mode)
            for name, env in MODES.items():
                for kk in KNOBS:
                    os.environ.pop(kk, None)
                os.environ.update(env)
                sc.zero_(); pa.zero_()
                if (seed + mode) % 2:
                    hs, hp = eng.run_device_through(mode, x, y, b.call_off, b.hdr, sc, pa, pinned=True, stream=torch.cuda.current_stream().cuda_stream)
                else:
                    eng.run_device(mode, x, y, b.call_off, b.hdr, sc, pa, stream=torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    hs, hp = sc.cpu().numpy(), pa.cpu().numpy()
                ok = np.array_equal(hs, ws) and np.array_equal(hp, wp) and np.array_equal(sc.cpu().numpy(), ws) and np.array_equal(pa.cpu().numpy(), wp)
                if not ok:
                    bad += 1
                    print(f"MISMATCH seed {seed} gmode {gmode} shape {shape} mode {mode} dispatch {name}", flush=True)
        print(f"seed {seed}: generator mode {gmode}, {ncalls} calls of {nmin}..{nmax} anchors ({b.nanchors} in all): "
              f"{'ok' if not bad else 'see above'}  [{time.time() - t0:.0f} s]", flush=True)
    eng.close()
    print("ALL IDENTICAL" if not bad else f"{bad} MISMATCHES")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
