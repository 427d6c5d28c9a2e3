#!/usr/bin/env python3
"""bench.py -- region-of-interest throughput of the GenArchBench hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bsw|...] [--items M]

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM (the reference's ROI likewise excludes file parsing).  The default
workload is the one BASELINE.json's metric is quoted on: bsw-large, 10 M pairs (configs[1]).
Ranks shard the item-id range with no collective on the data path: with N > 1 the FIXED large
input is split across the ranks (strong scaling, configs[4]; --scaling weak gives every rank a
full-size input of its own).

Output contract: the LAST line of stdout is ONE compact JSON object (< 4 KB: metric, value, unit,
n_gpus, steps, warmup, ms_per_step, scaling, dtype, config, roofline, cpu_baseline, parity and a
short `extra`).  Every other configuration of the suite is printed BEFORE it as a line of its own
({"suite": "<name>", ...}) and the full detail of everything goes to bench_suite.json next to this
file -- one grep-able timing line per run, as the reference's drivers print theirs
(bsw/src/main_banded.cpp:411-426, chain/src/main.cpp:201).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

# before anything starts the HIP runtime: pageable arrays are staged, not pinned in place (genarchbench_amd/__init__.py says why;
# nothing inside a timed region copies from pageable memory)
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1000000")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROUND = "r04"           # prefix of this round's evidence under profiles/
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


# HBM traffic of the dominant kernel(s) of a workload's LARGE configuration, measured once per round with rocprofv3 PMC
# counters in separate passes (tools/profiling/hbm_traffic.sh -> profiles/<ROUND>_hbm_traffic.json); `double_fetch`: the
# kernel reads wide coalesced streams, for which gfx950's FETCH_SIZE reports half the bytes (MI355X_MICROARCH.md, HBM)
TRAFFIC_KERNELS = {"bsw": (["bsw_dp8"], False), "chain": (["chain_block_kernel", "chain_facts_kernel", "ctab_geo<0", "ctab_fold<0"], False), "fast-chain": (["ctab_geo<1", "ctab_fold<1", "fastchain_kernel"], False),
                   "bpm": (["bpm_score32<", "bpm_score<"], False), "bitpal": (["bitpal_dp<true, true>"], False), "bitpal-edit": (["bitpal_edit_bv<"], False), "wfa": (["wfa_lds_static<16, false>"], False), "fmi": (["fmi_seed_kernel<true>"], False),
                   "fmi-sa": (["fmi_sa_kernel"], False), "parse-bsw": (["nl_count", "nl_fill", "bsw_meta", "bsw_codes", "len_offsets",
                                                                       "len_block_sums"], True)}


def pmc_traffic(name, is_large, kernel_ms):
    """-> (GB/s over the live kernel time, detail dict) or (None, None)"""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_hbm_traffic.json")      # this round's table or nothing: never a stale one
    if not is_large or name not in TRAFFIC_KERNELS or not os.path.exists(path) or not kernel_ms:
        return None, None
    tab = json.load(open(path)).get(name, {})
    kernels, double_fetch = TRAFFIC_KERNELS[name]
    rows = [v for name, v in tab.items() if any(name.startswith(k) for k in kernels)]      # names carry their template arguments
    f = sum(v.get("fetch_gb_raw", 0.0) for v in rows) * (2.0 if double_fetch else 1.0)
    w = sum(v.get("write_gb", 0.0) for v in rows)
    if f + w == 0:
        return None, None
    return round((f + w) / (kernel_ms * 1e-3), 3), {"fetch_GB_per_step": round(f, 3), "write_GB_per_step": round(w, 3),
                                                     "fetch_correction": "x2 (wide coalesced streams)" if double_fetch else "none (narrow or random accesses)",
                                                     "source": f"profiles/{ROUND}_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, one step"}


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup quota"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class omp_threads:
    """with omp_threads(n): OpenMP regions started from this thread use n threads (bench.py gives every rank of an N-GPU run
    cores / N threads; a step only one rank performs while the others wait may take them all)"""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        import ctypes as C
        try:
            self.omp = C.CDLL("libgomp.so.1")
            self.old = self.omp.omp_get_max_threads()
            self.omp.omp_set_num_threads(int(self.n))
        except OSError:
            self.omp = None

    def __exit__(self, *a):
        if self.omp is not None:
            self.omp.omp_set_num_threads(int(self.old))


# ---- the host-pointer path: what a drop-in driver times ("ROI on GPU includes H2D of packed inputs, kernels,
# and D2H of results").  Same scheme as benchmarks/common/gab_driver.h: the item range is cut into chunks, WORKERS host
# threads (each with its own engine handle = its own stream) pull chunk indices from a shared cursor and call the
# host-pointer entry point gab_*_run on page-locked slabs, so one chunk's copies run under another chunk's kernels.
HOST_WORKERS = int(os.environ.get("GAB_WORKERS_PER_GPU", "3"))


def pin(*arrays):
    """page-lock numpy arrays in place (gab_host_register), as the drivers do with their slabs before the ROI"""
    import ctypes as C
    from genarchbench_amd._lib import lib
    done = []
    for a in arrays:
        if a.nbytes and lib().gab_host_register(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)) == 0:
            done.append(a)
    return done


def unpin(arrays):
    import ctypes as C
    from genarchbench_amd._lib import lib
    for a in arrays:
        lib().gab_host_unregister(C.c_void_p(a.ctypes.data))


def host_queue(nchunks, make_state, run_chunk, close_state, passes=3):
    """-> MEAN wall time (s) of `passes` timed passes after one untimed pass (device buffers get allocated there), like
    the headline metric (a mean over the timed steps)"""
    import threading
    states = [make_state() for _ in range(max(1, min(HOST_WORKERS, nchunks)))]
    err = []

    def one_pass():
        cur = [0]
        lock = threading.Lock()

        def worker(st):
            try:
                while True:
                    with lock:
                        c = cur[0]; cur[0] += 1
                    if c >= nchunks:
                        return
                    run_chunk(st, c)            # a ctypes call: the GIL is released while it runs
            except Exception as e:              # noqa: BLE001
                err.append(e)
        th = [threading.Thread(target=worker, args=(st,)) for st in states]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0
    try:
        one_pass()
        mean = sum(one_pass() for _ in range(passes)) / passes
    finally:
        for st in states:
            close_state(st)
    if err:
        raise err[0]
    return mean


# ------------------------------------------------------------------------------------- bsw
class BswWorkload:
    name = "bsw"
    metric = "bsw ROI M alignments/sec"
    unit = "M alignments/s"
    dtype = "i16/i32"
    default_items = 10_000_000
    seed = 2

    def __init__(self, items, rank, dev, first=0, ids=None):
        import torch
        from tools import gabgen
        from genarchbench_amd.bsw import BandedPairWiseSW
        self.torch = torch
        self.items = items
        t0 = time.time()
        self.batch = gabgen.bsw(self.seed, items, 0, first=first)
        log(f"[rank {rank}] generated {items} bsw pairs in {time.time() - t0:.1f}s")
        b = self.batch
        t = lambda a: torch.from_numpy(a).to(dev)
        self.d = [t(b.ref), t(b.ref_off), t(b.qry), t(b.qry_off), t(b.len1), t(b.len2), t(b.h0)]
        self.score = torch.empty(items, dtype=torch.int32, device=dev)
        self.sw = BandedPairWiseSW(device=dev.index or 0)
        # algorithmic bytes per pair: len1 + len2 + 4 (h0) + 4 (score)  
        self.alg_bytes = int(b.len1.astype(np.int64).sum() + b.len2.astype(np.int64).sum() + 8 * items)
        self.kernel_ms = []
        self.cells = 0

    def step(self, stream):
        d = self.d
        self.sw.run_device(d[0], d[1], d[2], d[3], d[4], d[5], d[6], self.score, None, stream=stream)

    def after_step(self, timed):
        st = self.sw.last_stats()     # HIP events recorded on the launch stream around the DP kernels
        if timed:
            self.kernel_ms.append(st["kernel_ms"])
        self.cells = st["cells"]

    def check(self):
        """property check at full size + oracle check on a slice (the checker is not timed)"""
        from oracle import pyoracle
        from tools import gabgen
        got = self.score.cpu().numpy()
        b = self.batch
        assert (got >= b.h0).all(), "score below the seed score"
        assert (got <= b.h0 + b.len2).all(), "score above h0 + qlen*match"
        n = min(20000, self.items)
        sub = gabgen.BswBatch(b.ref, b.ref_off[:n], b.qry, b.qry_off[:n], b.len1[:n], b.len2[:n], b.h0[:n])
        want = pyoracle.bsw(sub)[:, 0]
        assert np.array_equal(got[:n], want), "bsw HIP output differs from the oracle"
        return f"bit-exact vs oracle on first {n} pairs; bounds hold on all {self.items}"

    def extra(self, ms_per_step):
        k = float(np.mean(self.kernel_ms)) if self.kernel_ms else None
        return {"gcups": round(self.cells / (ms_per_step * 1e6), 2) if ms_per_step else None,
                "cells_per_step": self.cells, "dominant_kernel": "bsw_dp8", "dominant_kernel_ms": k,
                "dominant_kernel_timing": "HIP events around the DP launches of a step: one bsw_dp8 launch per query-length class, "
                                          "two in flight at a time (two streams), so rocprofv3's per-launch durations overlap; the matching "
                                          "figure is the first-start-to-last-end span (tools/profiling/kernel_span.py on "
                                          f"profiles/{ROUND}_bsw_large_kernel_trace.csv)",
                # the bound that matters: integer VALU issue.  19.9 lane-instructions per DP cell is the PMC figure of the
                # end of r02 (SQ_INSTS_VALU x 64 / cells, profiles/r02_kernel_bounds.md).  Peak: MEASURED, profiles/r02_valu_issue.md
                # (tools/microbench/valu_issue.hip): the packed 16-bit, v_perm_b32, v_and_or_b32, v_lshl_or_b32, v_max3 and
                # 24-bit multiply instructions issue once per 4 cycles per SIMD whatever the occupancy (4.22 at the kernel's
                # ~2 waves per SIMD; 256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz / 4 = 39.3 T lane-instr/s), the non-packed 16-bit
                # max / sub, 32-bit add / and / shift-right / mov ones once per 2.56-2.81 cycles at 2 waves per SIMD.  The loop
                # is 33 instructions of the first and 29 of the second kind per four cells (disassembly), i.e. 3.5 cycles per
                # instruction on average = 45 T lane-instr/s for this mix at this occupancy (LDS-capped).
                "valu": {"lane_instr_per_cell": 19.9, "achieved_T_lane_instr_per_s":
                         round(19.9 * self.cells / (k * 1e9), 2) if k else None, "peak_T_lane_instr_per_s": 45.0,
                         "peak_source": "measured in round 2, profiles/r02_valu_issue.md (issue cost of the kernel's instruction classes at 2 waves "
                                        "per SIMD, weighted by the loop's mix: 33 x 4.22 + 29 x 2.6 cycles per 62 instructions)",
                         "frac": round(19.9 * self.cells / (k * 1e9) / 45.0, 3) if k else None,
                         "frac_if_every_instruction_cost_4_cycles": round(19.9 * self.cells / (k * 1e9) / 39.3, 3) if k else None}}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        ach = self.alg_bytes / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "bsw is integer-VALU/LDS bound by construction (~7.4k DP cells per ~210 input bytes); "
                        "see gcups in 'extra'"}

    def host_roi(self, chunk=1 << 20):
        import ctypes as C
        from genarchbench_amd._lib import check, lib
        from genarchbench_amd.bsw import BandedPairWiseSW
        b, n = self.batch, self.items
        score = np.full(n, -1, np.int32)
        pinned = pin(b.ref, b.qry, b.ref_off, b.qry_off, b.len1, b.len2, b.h0, score)
        at = lambda a, i: C.c_void_p(a.ctypes.data + a.itemsize * i)
        dv = self.sw.device

        def run(st, c):
            lo, hi = c * chunk, min(n, (c + 1) * chunk)
            check(lib().gab_bsw_run(st._h, at(b.ref, 0), at(b.ref_off, lo), at(b.qry, 0), at(b.qry_off, lo), at(b.len1, lo), at(b.len2, lo),
                                    at(b.h0, lo), C.c_int64(hi - lo), at(score, lo)))
        try:
            sec = host_queue((n + chunk - 1) // chunk, lambda: BandedPairWiseSW(device=dv), run, lambda st: st.close())
        finally:
            unpin(pinned)
        assert np.array_equal(score, self.score.cpu().numpy()), "host-pointer path and device path disagree"
        return {"ms": round(sec * 1e3, 3), "value": round(n / sec / 1e6, 3), "unit": self.unit, "chunk": chunk, "workers_per_gpu": HOST_WORKERS,
                "note": "gab_bsw_run on page-locked host slabs, chunks pulled by worker threads (the C driver's ROI): H2D + sort + DP + D2H; "
                        "all scores equal to the device path's"}

    def cpu_baseline(self, cores):
        """the compiled reference (oracle/_ref) if it travelled, else the oracle port; bounded sample"""
        from oracle import pyoracle
        from tools import gabgen
        flags = open("/proc/cpuinfo").read()
        isa = "avx512" if " avx512bw" in flags else "avx2"
        exe = pyoracle.ref_path("bsw_ref_" + isa)
        n = min(self.items, 4_000_000)
        if exe:
            with tempfile.TemporaryDirectory() as td:
                p = os.path.join(td, "bsw.txt")
                gabgen.write_text("bsw", p, self.seed, n, 0)
                env = dict(os.environ, OMP_PROC_BIND="true", OMP_PLACES="cores")
                secs = []
                for _ in range(3):          # the ROI of the sample is ~0.5 s and varies by 10-20 % between runs: three runs, median
                    r = subprocess.run([exe, "-pairs", p, "-t", str(cores), "-b", "512"], capture_output=True, text=True, env=env)
                    m = re.search(r"Overall SW cycles = (\d+)", r.stdout)
                    f = re.search(r"Processor freq: ([\d.]+) MHz", r.stdout)
                    if r.returncode != 0 or not m or not f or int(m.group(1)) <= 0:
                        break
                    secs.append(int(m.group(1)) / (float(f.group(1)) * 1e6))      # two decimals only in its "%0.2lf s": recompute
                if len(secs) == 3:
                    sec = sorted(secs)[1]
                    return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores,
                            "kind": "reference", "runs_M_per_s": [round(n / x / 1e6, 3) for x in secs],
                            "sample": f"first {n} pairs of the same seeded input, reference main_bsw ({isa} build) "
                                      f"-t {cores} -b 512, its own ROI timer: median of three runs ({sec:.2f} s; all three in runs_M_per_s)"}
                log("reference binary failed, falling back to the oracle port:", r.stderr[-300:])
        n = min(self.items, 400_000)
        b = self.batch
        sub = gabgen.BswBatch(b.ref, b.ref_off[:n], b.qry, b.qry_off[:n], b.len1[:n], b.len2[:n], b.h0[:n])
        t0 = time.time()
        pyoracle.bsw(sub, threads=cores)
        sec = time.time() - t0
        return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"first {n} pairs, oracle/bsw.c scalar restatement + OpenMP ({sec:.2f} s)"}


# ------------------------------------------------------------------------- chain / fast-chain
class ChainWorkload:
    name = "chain"
    mode = 0
    metric = "chain ROI M seeds/sec"
    unit = "M seeds/s"
    dtype = "i32+f64"
    default_items = 10_000          # calls per GPU (chain-large: c_elegans 10k calls)
    seed = 5
    ref_exe = "chain_ref"
    # r04: calls of >= 16 000 anchors run in the table form (chain_tab.hip: geometry + fold), the rest in chain_block_kernel beside them
    kernel = "chain_block_kernel (calls < 16000 anchors), ctab_geo<0, false> + ctab_fold<0> (the rest)"

    def __init__(self, items, rank, dev, first=0, ids=None):
        import torch
        from tools import gabgen
        from genarchbench_amd.chain import ChainEngine
        self.torch = torch
        self.calls = items
        t0 = time.time()
        # ids: this rank's calls of the fixed input under strong scaling (dealt longest first, genarchbench_amd/shard.py)
        self.batch = b = (gabgen.chain(self.seed, items, 0, 50, 60000, first=first) if ids is None else
                          gabgen.chain_ids(self.seed, ids, 0, 50, 60000))
        self.items = b.nanchors       # the metric counts seeds (anchors)
        log(f"[rank {rank}] generated {items} calls / {b.nanchors} anchors in {time.time() - t0:.1f}s")
        self.x = torch.from_numpy(b.x.view(np.int64)).to(dev)
        self.y = torch.from_numpy(b.y.view(np.int64)).to(dev)
        self.score = torch.empty(b.nanchors, dtype=torch.int32, device=dev)
        self.parent = torch.empty(b.nanchors, dtype=torch.int32, device=dev)
        self.dev_index = dev.index or 0
        self.eng = ChainEngine(device=self.dev_index)
        self.alg_bytes = 24 * b.nanchors + 24 * items      # 24 B per seed + header
        self.kernel_ms = []
        self.evals = 0

    def step(self, stream):
        self.eng.run_device(self.mode, self.x, self.y, self.batch.call_off, self.batch.hdr, self.score,
                            self.parent, stream=stream)

    def after_step(self, timed):
        st = self.eng.last_stats()
        if timed:
            self.kernel_ms.append(st["kernel_ms"])
        self.evals = st["evals"]

    def check(self):
        from oracle import pyoracle
        from tools import gabgen
        b = self.batch
        got_s = self.score.cpu().numpy(); got_p = self.parent.cpu().numpy()
        # size-independent properties: parent precedes child, scores >= own q_span
        idx = np.arange(b.nanchors, dtype=np.int64) - np.repeat(b.call_off, b.hdr["n"])
        assert (got_p < idx).all() and (got_p >= -1).all(), "parent index out of range"
        assert (got_s >= ((b.y >> np.uint64(32)) & np.uint64(0xff)).astype(np.int32)).all(), "score below q_span"
        c = min(300, self.calls)
        end = int(b.call_off[c - 1] + b.hdr["n"][c - 1])
        sub = gabgen.ChainBatch(b.hdr[:c], b.call_off[:c], b.x[:end], b.y[:end])
        ws, wp = pyoracle.chain(sub, self.mode)
        assert np.array_equal(got_s[:end], ws) and np.array_equal(got_p[:end], wp), "chain HIP output differs from the oracle"
        return f"bit-exact vs oracle on first {c} calls ({end} seeds); structural checks on all {b.nanchors}"

    def extra(self, ms_per_step):
        k = float(np.mean(self.kernel_ms)) if self.kernel_ms else None
        return {"calls": self.calls, "seeds": self.items, "pred_evals_per_step": self.evals,
                "g_evals_per_s": round(self.evals / (ms_per_step * 1e6), 3) if ms_per_step else None,
                "dominant_kernel": self.kernel, "dominant_kernel_ms": k}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        ach = self.alg_bytes / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "24 B/seed of HBM traffic vs ~100-200 predecessor evaluations/seed served from "
                        "registers/L1/L2: latency+VALU bound by construction; see g_evals_per_s"}

    def host_roi(self):
        import ctypes as C
        from genarchbench_amd._lib import check, lib
        from genarchbench_amd.chain import ChainEngine
        b = self.batch
        score = np.full(b.nanchors, -7, np.int32); parent = np.full(b.nanchors, -7, np.int32)
        pinned = pin(b.x, b.y, score, parent)
        at = lambda a, i: C.c_void_p(a.ctypes.data + a.itemsize * i)
        # chunks of calls as benchmarks/chain/src/main.c cuts them: ~equal anchor counts, offsets re-based to the window
        nchunks = int(os.environ.get("GAB_CHAIN_HOST_CHUNKS", "1"))
        per = b.nanchors // nchunks + 1
        beg, acc = [0], 0
        for c in range(self.calls):
            acc += int(b.hdr["n"][c])
            if acc >= per and len(beg) < nchunks:
                beg.append(c + 1); acc = 0
        beg.append(self.calls)
        offs = [np.ascontiguousarray(b.call_off[beg[k]:beg[k + 1]] - b.call_off[beg[k]]) for k in range(len(beg) - 1)]
        dv = self.dev_index

        def run(st, k):
            lo, hi = beg[k], beg[k + 1]
            if hi <= lo:
                return
            a0 = int(b.call_off[lo])
            check(lib().gab_chain_run(st._h, C.c_int(self.mode), at(b.x, a0), at(b.y, a0), at(offs[k], 0), at(b.hdr, lo), C.c_int64(hi - lo),
                                      at(score, a0), at(parent, a0)))
        try:
            sec = host_queue(len(beg) - 1, lambda: ChainEngine(device=dv), run, lambda st: st.close())
        finally:
            unpin(pinned)
        assert np.array_equal(score, self.score.cpu().numpy()) and np.array_equal(parent, self.parent.cpu().numpy()), \
            "host-pointer path and device path disagree"
        return {"ms": round(sec * 1e3, 3), "value": round(b.nanchors / sec / 1e6, 3), "unit": self.unit, "chunks": len(beg) - 1,
                "workers_per_gpu": HOST_WORKERS,
                "note": "gab_chain_run on page-locked host arrays (the C driver's ROI): a kernel fetches x, y longest call first, the DP "
                        "workgroups wait per call and write scores, parents through; "
                        "all results equal to the device path's"}

    def cpu_baseline(self, cores):
        from oracle import pyoracle
        from tools import gabgen
        c = min(self.calls, 1500)
        exe = pyoracle.ref_path(self.ref_exe if self.mode == 0 else
                                ("fastchain_ref_avx512" if " avx512bw" in open("/proc/cpuinfo").read() else "fastchain_ref_avx2"))
        b = self.batch
        end = int(b.call_off[c - 1] + b.hdr["n"][c - 1])
        if exe:
            with tempfile.TemporaryDirectory() as td:
                p = os.path.join(td, "chain.txt")
                gabgen.write_text("chain", p, self.seed, c, 0, 50, 60000)
                env = dict(os.environ, OMP_PROC_BIND="true", OMP_PLACES="cores")
                r = subprocess.run([exe, "-i", p, "-o", os.path.join(td, "out.txt"), "-t", str(cores)],
                                   capture_output=True, text=True, env=env)
                m = re.search(r"Time in kernel: ([\d.]+) sec", r.stderr)
                if r.returncode == 0 and m and float(m.group(1)) > 0:
                    sec = float(m.group(1))
                    return {"value": round(end / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "reference",
                            "sample": f"first {c} calls ({end} seeds) of the same seeded input, reference "
                                      f"{os.path.basename(exe)} -t {cores}, its own ROI timer ({sec:.2f} s)"}
                log("reference binary failed or ROI too short, using the oracle port:", r.stderr[-200:])
        sub = gabgen.ChainBatch(b.hdr[:c], b.call_off[:c], b.x[:end], b.y[:end])
        t0 = time.time()
        pyoracle.chain(sub, self.mode, threads=cores)
        sec = time.time() - t0
        return {"value": round(end / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"first {c} calls ({end} seeds), oracle/chain.c + OpenMP dynamic ({sec:.2f} s)"}


class FastChainWorkload(ChainWorkload):
    name = "fast-chain"
    mode = 1
    metric = "fast-chain ROI M seeds/sec"
    dtype = "i32+f32"
    # r04: calls of >= 4 096 anchors run in the table form whatever the batch (chain_tab.hip: geometry + fold), the rest in
    # fastchain_kernel beside them; the event-timed region covers all of them
    kernel = "ctab_geo<1, false> + ctab_fold<1> (calls >= 4096 anchors), fastchain_kernel (the rest)"


# ------------------------------------------------------------------------------------- bpm
class BpmWorkload:
    name = "bpm"
    metric = "bpm ROI M alignments/sec"
    unit = "M alignments/s"
    dtype = "u64"
    default_items = 10_000_000
    seed = 3
    plen = 151

    def __init__(self, items, rank, dev, first=0, ids=None):
        import torch
        from tools import gabgen
        from genarchbench_amd.bpm import BpmEngine
        self.items = items
        t0 = time.time()
        raw = gabgen.pairs(self.seed, items, 0, self.plen, first=first)
        self.batch = b = raw.swapped_combined()      # the driver's longer-is-pattern swap
        log(f"[rank {rank}] generated {items} bpm pairs in {time.time() - t0:.1f}s")
        t = lambda a: torch.from_numpy(a).to(dev)
        slab = t(b.pat)
        self.d = [slab, t(b.pat_off), t(b.pat_len), slab, t(b.txt_off), t(b.txt_len)]
        self.score = torch.empty(items, dtype=torch.int32, device=dev)
        self.dev_index = dev.index or 0
        self.eng = BpmEngine(device=self.dev_index)
        self.alg_bytes = int(b.pat_len.astype(np.int64).sum() + b.txt_len.astype(np.int64).sum() + 4 * items)
        self.kernel_ms, self.total_ms = [], []
        self.stats = {}

    def step(self, stream):
        d = self.d
        self.eng.run_device(d[0], d[1], d[2], d[3], d[4], d[5], self.score, stream=stream)

    def after_step(self, timed):
        st = self.eng.last_stats()
        if timed:
            self.kernel_ms.append(st["kernel_ms"]); self.total_ms.append(st["total_ms"])
        self.stats = st

    def check(self):
        from oracle import pyoracle
        from tools import gabgen
        got = self.score.cpu().numpy()
        b = self.batch
        assert (got <= 0).all() and (got >= -b.pat_len).all(), "score outside [-plen, 0]"
        assert (-got >= b.pat_len - b.txt_len).all(), "distance below the length difference"
        n = min(50000, self.items)
        sub = gabgen.PairBatch(b.pat, b.pat_off[:n], b.pat_len[:n], b.txt, b.txt_off[:n], b.txt_len[:n])
        assert np.array_equal(got[:n], pyoracle.bpm(sub)), "bpm HIP output differs from the oracle"
        return f"bit-exact vs oracle on first {n} pairs; bounds hold on all {self.items}"

    def extra(self, ms_per_step):
        return {"block_steps_per_step": self.stats.get("block_steps"), "full_path_pairs": self.stats.get("full_pairs"),
                "g_block_steps_per_s": round(self.stats.get("block_steps", 0) / (ms_per_step * 1e6), 2),
                "dominant_kernel": "bpm_score32<5> (+ bpm_band<5> of the previous slice on a second stream)",
                "dominant_kernel_ms": float(np.mean(self.kernel_ms)),
                "device_total_ms": float(np.mean(self.total_ms))}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        ach = self.alg_bytes / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "score kernel: plen+tlen+4 B per pair vs ~9.4k integer VALU per pair (61 per text base: the column as five 32-bit words; VALU bound)"}

    host_entry, host_reserve = "gab_bpm_run", "gab_bpm_reserve"

    def _host_engine(self):
        from genarchbench_amd.bpm import BpmEngine
        return BpmEngine(device=self.dev_index)

    def host_roi(self, chunk=1 << 20):
        """the C driver's ROI (bpm/tools/align_benchmark.c:213-337 in the reference): host slabs in, scores out"""
        import ctypes as C
        from genarchbench_amd._lib import check, lib
        n = self.items
        b = self.batch.interleaved()       # the layout of the driver's slab (the pair file): pattern i, text i, pattern i + 1, ...
        score = np.full(n, 12345, np.int32)
        pinned = pin(b.pat, b.pat_off, b.pat_len, b.txt_off, b.txt_len, score)       # b.txt IS b.pat
        at = lambda a, i: C.c_void_p(a.ctypes.data + a.itemsize * i)
        entry, reserve = self.host_entry, self.host_reserve
        slab_bytes = int(b.pat.nbytes)

        def make():
            e = self._host_engine()
            # buffers for the largest chunk + warm copy queues, outside the ROI (the drivers do the same)
            check(getattr(lib(), reserve)(e._h, C.c_int64(min(n, chunk)), C.c_int64(min(slab_bytes, 2 * 160 * chunk))))
            return e

        def run(st, c):
            lo, hi = c * chunk, min(n, (c + 1) * chunk)
            check(getattr(lib(), entry)(st._h, at(b.pat, 0), at(b.pat_off, lo), at(b.pat_len, lo), at(b.txt, 0), at(b.txt_off, lo),
                                         at(b.txt_len, lo), C.c_int64(hi - lo), at(score, lo)))
        try:
            sec = host_queue((n + chunk - 1) // chunk, make, run, lambda st: st.close())
        finally:
            unpin(pinned)
        assert np.array_equal(score, self.score.cpu().numpy()), "host-pointer path and device path disagree"
        return {"ms": round(sec * 1e3, 3), "value": round(n / sec / 1e6, 3), "unit": self.unit, "chunk": chunk, "workers_per_gpu": HOST_WORKERS,
                "note": f"{entry} on page-locked host slabs, chunks pulled by worker threads (the C driver's ROI): the pair text over the "
                        "bus (~310 B per pair) + kernels + the scores back; all scores equal to the device path's"}

    def cpu_baseline(self, cores):
        from oracle import pyoracle
        from tools import gabgen
        exe = pyoracle.ref_path("bpm_ref")
        n = min(self.items, 2_000_000)
        if exe:
            with tempfile.TemporaryDirectory() as td:
                p = os.path.join(td, "bpm.txt")
                gabgen.write_text("bpm", p, self.seed, n, 0, self.plen)
                env = dict(os.environ, OMP_PROC_BIND="true", OMP_PLACES="cores")
                r = subprocess.run([exe, "-a", "bpm-edit", "-i", p, "-t", str(cores)], capture_output=True, text=True, env=env)
                m = re.search(r"Time.Benchmark\s+([\d.]+) (ms|s|us)", r.stderr)
                if r.returncode == 0 and m:
                    sec = float(m.group(1)) * {"s": 1.0, "ms": 1e-3, "us": 1e-6}[m.group(2)]
                    return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "reference",
                            "sample": f"first {n} pairs of the same seeded input, reference align_benchmark -a bpm-edit "
                                      f"-t {cores}, its own Time.Benchmark ({sec:.2f} s)"}
                log("reference binary failed, using the oracle port:", r.stderr[-200:])
        b = self.batch
        n = min(self.items, 1_000_000)
        sub = gabgen.PairBatch(b.pat, b.pat_off[:n], b.pat_len[:n], b.txt, b.txt_off[:n], b.txt_len[:n])
        t0 = time.time(); pyoracle.bpm(sub, threads=cores); sec = time.time() - t0
        return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"first {n} pairs, oracle/bpm.c + OpenMP ({sec:.2f} s)"}


# ------------------------------------------------------------------------------------- bitpal (bpm -a bitpal-scored / bitpal-edit)
class BitpalWorkload(BpmWorkload):
    name = "bitpal"
    algorithm, alg_name = 1, "bitpal-scored"
    metric = "bpm (bitpal-scored) ROI M alignments/sec"
    dtype = "i32"

    def __init__(self, items, rank, dev, first=0, ids=None):
        import torch
        from tools import gabgen
        from genarchbench_amd.bitpal import BitpalEngine
        self.items = items
        t0 = time.time()
        raw = gabgen.pairs(self.seed, items, 0, self.plen, first=first)
        self.batch = b = raw.swapped_combined()
        log(f"[rank {rank}] generated {items} bpm pairs in {time.time() - t0:.1f}s")
        t = lambda a: torch.from_numpy(a).to(dev)
        slab = t(b.pat)
        self.d = [slab, t(b.pat_off), t(b.pat_len), slab, t(b.txt_off), t(b.txt_len)]
        self.score = torch.empty(items, dtype=torch.int32, device=dev)
        self.dev_index = dev.index or 0
        self.eng = BitpalEngine(self.algorithm, device=self.dev_index)
        self.alg_bytes = int(b.pat_len.astype(np.int64).sum() + b.txt_len.astype(np.int64).sum() + 4 * items)
        self.kernel_ms, self.total_ms = [], []
        self.stats = {}

    host_entry, host_reserve = "gab_bitpal_run", "gab_bitpal_reserve"

    def _host_engine(self):
        from genarchbench_amd.bitpal import BitpalEngine
        return BitpalEngine(self.algorithm, device=self.dev_index)

    def check(self):
        from oracle import pyoracle
        from tools import gabgen
        got = self.score.cpu().numpy()
        b = self.batch
        gap = -2 if self.algorithm else -1
        assert (got >= gap * (b.pat_len + b.txt_len)).all() and (got <= self.algorithm * np.minimum(b.pat_len, b.txt_len)).all(), "score out of bounds"
        n = min(50000, self.items)
        sub = gabgen.PairBatch(b.pat, b.pat_off[:n], b.pat_len[:n], b.txt, b.txt_off[:n], b.txt_len[:n])
        assert np.array_equal(got[:n], pyoracle.bitpal(sub, self.algorithm)), "bitpal HIP output differs from the oracle"
        return f"bit-exact vs oracle on first {n} pairs; bounds hold on all {self.items}"

    def extra(self, ms_per_step):
        k = float(np.mean(self.kernel_ms))
        return {"dp_cells_per_step": self.stats.get("cells"), "long_pairs": self.stats.get("long_pairs"),
                "gcups_kernel": round(self.stats.get("cells", 0) / (k * 1e6), 1),
                # 3.5 VALU per DP cell in the ISA of bitpal_dp (2 per cell + 6 per four columns) against 256 CUs x 4 SIMDs
                # issuing one wave64 VALU instruction every 4 cycles at 2.4 GHz
                "valu_issue_frac_est": round(self.stats.get("cells", 0) * 3.5 / 64 / (k * 1e-3) / (256 * 2.4e9), 3),
                "dominant_kernel": "bitpal_dp", "dominant_kernel_ms": k, "device_total_ms": float(np.mean(self.total_ms))}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        ach = self.alg_bytes / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "plen+tlen+4 B per pair vs 3.5 integer VALU per DP cell (22.8 k cells per 151-bp pair): VALU-issue bound, see extra.valu_issue_frac_est"}

    def cpu_baseline(self, cores):
        from oracle import pyoracle
        from tools import gabgen
        exe = pyoracle.ref_path("bpm_ref")
        n = min(self.items, 500_000)
        if exe:
            with tempfile.TemporaryDirectory() as td:
                p = os.path.join(td, "bpm.txt")
                gabgen.write_text("bpm", p, self.seed, n, 0, self.plen)
                env = dict(os.environ, OMP_PROC_BIND="true", OMP_PLACES="cores")
                r = subprocess.run([exe, "-a", self.alg_name, "-i", p, "-t", str(cores)], capture_output=True, text=True, env=env)
                m = re.search(r"Time.Benchmark\s+([\d.]+) (ms|s|us)", r.stderr)
                if r.returncode == 0 and m:
                    sec = float(m.group(1)) * {"s": 1.0, "ms": 1e-3, "us": 1e-6}[m.group(2)]
                    return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "reference",
                            "sample": f"first {n} pairs of the same seeded input, reference align_benchmark -a {self.alg_name} "
                                      f"-t {cores}, its own Time.Benchmark ({sec:.2f} s)"}
                log("reference binary failed, using the oracle port:", r.stderr[-200:])
        b = self.batch
        n = min(self.items, 200_000)
        sub = gabgen.PairBatch(b.pat, b.pat_off[:n], b.pat_len[:n], b.txt, b.txt_off[:n], b.txt_len[:n])
        t0 = time.time(); pyoracle.bitpal(sub, self.algorithm, threads=cores); sec = time.time() - t0
        return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"first {n} pairs, oracle/bitpal.c + OpenMP ({sec:.2f} s)"}


class BitpalEditWorkload(BitpalWorkload):
    name = "bitpal-edit"
    algorithm, alg_name = 0, "bitpal-edit"
    metric = "bpm (bitpal-edit) ROI M alignments/sec"

    def extra(self, ms_per_step):
        # -a bitpal-edit is minus the edit distance: it runs as Myers' bit-vector (bitpal_edit_bv<D>, gab_bitvec.h: ~60 VALU
        # instructions per COLUMN of a 151-row pair), the integer DP (bitpal_dp, 3.5 per CELL) only takes its rejects
        k = float(np.mean(self.kernel_ms))
        return {"dp_cells_per_step": self.stats.get("cells"), "long_pairs": self.stats.get("long_pairs"),
                "gcups_kernel": round(self.stats.get("cells", 0) / (k * 1e6), 1),
                "dominant_kernel": "bitpal_edit_bv", "dominant_kernel_ms": k, "device_total_ms": float(np.mean(self.total_ms))}

    def roofline(self):
        r = super().roofline()
        r["note"] = ("plen+tlen+4 B per pair vs ~60 VALU per text base and pair (Myers' bit-vector, five 32-bit words for 151 rows): "
                     "VALU-issue bound")
        return r


# ------------------------------------------------------------------------------------- wfa
class WfaWorkload:
    name = "wfa"
    metric = "wfa ROI M alignments/sec"
    unit = "M alignments/s"
    dtype = "i16"
    default_items = 1_000_000
    seed = 4
    plen = 151

    def __init__(self, items, rank, dev, first=0, ids=None):
        import torch
        from tools import gabgen
        from genarchbench_amd.wfa import AffineWavefronts, ops_layout
        self.items = items
        t0 = time.time()
        self.batch = b = gabgen.pairs(self.seed, items, 0, self.plen, first=first)
        log(f"[rank {rank}] generated {items} wfa pairs in {time.time() - t0:.1f}s")
        t = lambda a: torch.from_numpy(a).to(dev)
        self.off, total = ops_layout(b)
        self.d = [t(b.pat), t(b.pat_off), t(b.pat_len), t(b.txt), t(b.txt_off), t(b.txt_len)]
        self.ops = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
        self.d_off = t(self.off)
        self.ops_len = torch.zeros(items, dtype=torch.int32, device=dev)
        self.score = torch.zeros(items, dtype=torch.int32, device=dev)
        self.dev_index = dev.index or 0
        self.eng = AffineWavefronts(device=self.dev_index)
        self.in_bytes = int(b.pat_len.astype(np.int64).sum() + b.txt_len.astype(np.int64).sum() + 4 * items)
        self.alg_bytes = self.in_bytes
        self.kernel_ms, self.total_ms = [], []
        self.stats = {}

    def step(self, stream):
        d = self.d
        self.eng.run_device(d[0], d[1], d[2], d[3], d[4], d[5], self.ops, self.d_off, self.ops_len, self.score, stream=stream)

    def after_step(self, timed):
        st = self.eng.last_stats()
        if timed:
            self.kernel_ms.append(st["kernel_ms"]); self.total_ms.append(st["total_ms"])
        self.stats = st

    def check(self):
        from oracle import pyoracle
        from tools import gabgen
        b = self.batch
        ln = self.ops_len.cpu().numpy(); sc = self.score.cpu().numpy()
        self.alg_bytes = self.in_bytes + int(ln.astype(np.int64).sum())     # + cigar bytes
        assert (ln >= np.maximum(b.pat_len, b.txt_len)).all() and (ln <= b.pat_len + b.txt_len).all(), "cigar length out of range"
        assert (sc >= 0).all()
        n = min(20000, self.items)
        sub = gabgen.PairBatch(b.pat, b.pat_off[:n], b.pat_len[:n], b.txt, b.txt_off[:n], b.txt_len[:n])
        wo, woff, wl, ws = pyoracle.wfa(sub)
        assert np.array_equal(sc[:n], ws) and np.array_equal(ln[:n], wl), "wfa score / cigar length differ from the oracle"
        end = int(woff[n - 1] + b.pat_len[n - 1] + b.txt_len[n - 1])
        go = self.ops[:end].cpu().numpy()
        for i in range(0, n, 3):
            assert np.array_equal(go[woff[i]:woff[i] + wl[i]], wo[woff[i]:woff[i] + wl[i]]), f"cigar {i} differs"
        return f"bit-exact CIGARs vs oracle on {n // 3} of the first {n} pairs (+ all their scores); bounds on all {self.items}"

    def extra(self, ms_per_step):
        return {"work_units_per_step": self.stats.get("work"), "requeued_pairs": self.stats.get("requeued"),
                "dominant_kernel": "wfa_lds_static<16, false> (first launch)", "dominant_kernel_ms": float(np.mean(self.kernel_ms)),
                "device_total_ms": float(np.mean(self.total_ms))}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        ach = self.alg_bytes / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "plen+tlen+cigar+4 B per pair; the kernel is instruction-issue bound (83 % VALU busy at 45 % lane utilisation: four pairs per wave, lanes = diagonals)"}

    def host_roi(self, chunk=1 << 18):
        """the C driver's ROI (wfa/tools/align_benchmark.c:378-491 in the reference): pair text in, the alignments out -- as the
        run-length CIGAR text the driver prints (gab_wfa_run_packed), not as pattern + text length bytes of operation room"""
        import ctypes as C
        from genarchbench_amd._lib import check, lib
        from genarchbench_amd.wfa import AffineWavefronts
        n = self.items
        b = self.batch.interleaved()       # the layout of the driver's slab (the pair file): pattern i, text i, pattern i + 1, ...
        nchunks = (n + chunk - 1) // chunk
        room = (b.pat_len.astype(np.int64) + b.txt_len.astype(np.int64))
        beg = np.zeros(nchunks + 1, np.int64)                    # text room of chunk c: a quarter of its operation room
        for c in range(nchunks):
            beg[c + 1] = beg[c] + ((int(room[c * chunk:(c + 1) * chunk].sum()) // 4 + 4096 + 255) & ~255)
        text = np.zeros(int(beg[-1]) + 16, np.uint8)
        off = np.full(n, -1, np.int64); ln = np.full(n, -1, np.int32); sc = np.full(n, -1, np.int32)
        used = np.zeros(nchunks, np.int64)
        pinned = pin(b.pat, b.pat_off, b.txt_off, b.pat_len, b.txt_len, text, off, ln, sc)
        at = lambda a, i: C.c_void_p(a.ctypes.data + a.itemsize * i)
        stride = (int(room.max()) + 7) & ~7

        def make():
            e = AffineWavefronts(device=self.dev_index)
            check(lib().gab_wfa_reserve(e._h, C.c_int64(min(n, chunk)), C.c_int64(int(b.pat.nbytes) // max(nchunks, 1) + (1 << 20)),
                                        C.c_int64(stride * min(n, chunk) + int(np.diff(beg).max()) + 4096)))
            return e

        def run(st, c):
            lo, hi = c * chunk, min(n, (c + 1) * chunk)
            need = C.c_int64(0)
            check(lib().gab_wfa_run_packed(st._h, at(b.pat, 0), at(b.pat_off, lo), at(b.pat_len, lo), at(b.txt, 0), at(b.txt_off, lo), at(b.txt_len, lo),
                                           C.c_int64(hi - lo), at(text, int(beg[c])), C.c_int64(int(beg[c + 1] - beg[c])), at(off, lo), at(ln, lo), at(sc, lo),
                                           C.byref(need)))
            used[c] = need.value
        try:
            sec = host_queue(nchunks, make, run, lambda st: st.close())
        finally:
            unpin(pinned)
        dl = self.ops_len.cpu().numpy()
        assert np.array_equal(sc, self.score.cpu().numpy()), "host-pointer path and device path disagree"
        dev_ops = self.ops.cpu().numpy()
        for i in range(0, n, max(1, n // 5000)):                 # the text is the run-length form of the device path's operations
            o = int(self.off[i]); ops = dev_ops[o:o + dl[i]]
            cut = np.flatnonzero(np.diff(ops)) + 1
            runs = np.diff(np.concatenate(([0], cut, [len(ops)])))
            want = b"".join(b"%d%c" % (r, ops[k]) for r, k in zip(runs, np.concatenate(([0], cut)))) if len(ops) else b""
            t0 = int(beg[i // chunk] + off[i])
            assert text[t0:t0 + ln[i]].tobytes() == want, f"cigar {i}: host-pointer path and device path disagree"
        return {"ms": round(sec * 1e3, 3), "value": round(n / sec / 1e6, 3), "unit": self.unit, "chunk": chunk, "workers_per_gpu": HOST_WORKERS,
                "bytes_in": int(b.pat.nbytes + 24 * n), "bytes_out": int(used.sum() + 16 * n),
                "note": "gab_wfa_run_packed on page-locked host slabs, chunks pulled by worker threads (the C driver's ROI): the sequences over the "
                        "bus + kernels + the printed CIGAR text back (run-length encoded on the device); all scores and a sample of the "
                        "CIGARs equal to the device path's"}

    def cpu_baseline(self, cores):
        from oracle import pyoracle
        from tools import gabgen
        exe = pyoracle.ref_path("wfa_ref")
        n = min(self.items, 1_000_000)
        if exe:
            with tempfile.TemporaryDirectory() as td:
                p = os.path.join(td, "wfa.txt")
                gabgen.write_text("wfa", p, self.seed, n, 0, self.plen)
                env = dict(os.environ, OMP_PROC_BIND="true", OMP_PLACES="cores")
                r = subprocess.run([exe, "-i", p, "-t", str(cores)], capture_output=True, text=True, env=env)
                m = re.search(r"Time.Alignment:\s+([\d.]+) s", r.stdout)
                if r.returncode == 0 and m and float(m.group(1)) > 0:
                    sec = float(m.group(1))
                    return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "reference",
                            "sample": f"first {n} pairs of the same seeded input, reference align_benchmark -t {cores}, "
                                      f"its own Time.Alignment ({sec:.3f} s)"}
                log("reference binary failed, using the oracle port:", r.stderr[-200:])
        b = self.batch
        n = min(self.items, 300_000)
        sub = gabgen.PairBatch(b.pat, b.pat_off[:n], b.pat_len[:n], b.txt, b.txt_off[:n], b.txt_len[:n])
        t0 = time.time(); pyoracle.wfa(sub, threads=cores); sec = time.time() - t0
        return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"first {n} pairs, oracle/wfa.c + OpenMP ({sec:.2f} s)"}


# ------------------------------------------------------------------------------------- fmi
class FmiWorkload:
    name = "fmi"
    metric = "fmi ROI M reads/sec"
    unit = "M reads/s"
    dtype = "i64"
    default_items = 10_000_000
    seed = 6
    ref_mbp = int(os.environ.get("GAB_FMI_REF_MBP", "256"))   # synthetic reference size (>= 256 Mbp)
    readlen = 151
    min_seed_len = 19
    wide_lists = False

    def __init__(self, items, rank, dev, first=0, ids=None):
        import torch
        from tools import gabgen, mkindex
        from genarchbench_amd.fmi import FMI_search
        self.items = items
        t0 = time.time()
        key = (self.seed, self.ref_mbp)
        world, dist = _CTX.get("world", 1), _CTX.get("dist")
        if key not in _FMI_INDEX_CACHE:                 # fmi, fmi (16-byte lists) and fmi-sa of one run share the index
            _FMI_INDEX_CACHE.clear()
            share = os.path.join(_CTX["share_dir"], f"fmi_{self.seed}_{self.ref_mbp}") if world > 1 else None
            if world == 1 or rank == 0:
                ref = gabgen.fmi_ref(self.seed, self.ref_mbp * 1_000_000, 5)
                log(f"[rank {rank}] generated the {self.ref_mbp} Mbp reference in {time.time() - t0:.1f}s")
                t0 = time.time()
                with omp_threads(host_cores()):              # the other ranks wait at the barrier below: all host cores for the build
                    index = mkindex.FmIndex(ref)             # outside the ROI, like load_index in the reference
                log(f"[rank {rank}] built the FM-index ({index.ref_seq_len} rows, {len(index.cp_occ) / 2**20:.0f} MiB of CP_OCC) "
                    f"in {time.time() - t0:.1f}s")
                if share:                                    # ONE index per node: the other ranks map rank 0's file, as the
                    index.write(share)                       # reference's threads share one FMI_search (fmi/fmi.cpp:102-103)
                    np.save(share + ".ref.npy", ref)
            if world > 1:
                dist.barrier()
                if rank != 0:
                    ref = np.load(share + ".ref.npy", mmap_mode="r")
                    index = mkindex.IndexFile(share)
                    log(f"[rank {rank}] mapped rank 0's index file ({index.ref_seq_len} rows)")
            _FMI_INDEX_CACHE[key] = (ref, index)
        self.ref, self.index = _FMI_INDEX_CACHE[key]
        self.reads = gabgen.fmi_reads(self.seed + 1, np.ascontiguousarray(self.ref), items, self.readlen, self.readlen, first=first)
        if self.wide_lists:
            os.environ["GAB_FMI_WIDE_LISTS"] = "1"      # read when the handle is made
        try:
            self.eng = FMI_search(arrays=(self.index.ref_seq_len, self.index.count, self.index.cp_occ, self.index.sentinel_index),
                                  device=dev.index or 0)
        finally:
            os.environ.pop("GAB_FMI_WIDE_LISTS", None)
        self.dev_index = dev.index or 0
        self.enc = torch.from_numpy(self.reads.enc).to(dev)
        self.len = torch.from_numpy(self.reads.len).to(dev)
        self.kernel_ms = []
        self.stats = {}
        self.result = None

    def step(self, stream):
        self.result = self.eng.seed_device(self.enc, self.len, self.min_seed_len, stream=stream)

    def after_step(self, timed):
        st = self.eng.last_stats()
        if timed:
            self.kernel_ms.append(st["kernel_ms"])
        self.stats = st

    def check(self):
        import ctypes as C
        from oracle import pyoracle
        from tools import gabgen
        from genarchbench_amd.fmi import SMEM_DTYPE
        d_out, d_off, n = self.result
        hip = C.CDLL("libamdhip64.so")
        off = np.zeros(self.items + 1, np.int64)
        assert hip.hipMemcpy(off.ctypes.data_as(C.c_void_p), C.c_void_p(d_off), C.c_size_t(8 * (self.items + 1)), C.c_int(2)) == 0
        assert off[-1] == n and (np.diff(off) >= 0).all(), "read offsets are not a prefix sum"
        nr = min(20000, self.items)
        k = int(off[nr])
        host = np.zeros(max(k, 1) * 40, np.uint8)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(d_out), C.c_size_t(k * 40), C.c_int(2)) == 0
        got = host[:k * 40].view(SMEM_DTYPE)
        oidx = pyoracle.FmIndex()
        cnt = (C.c_int64 * 5)(*[int(x) for x in self.index.count])
        pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(self.index.ref_seq_len), cnt,
                                              self.index.cp_occ.ctypes.data_as(C.c_void_p), C.c_int64(self.index.sentinel_index))
        sub = gabgen.ReadBatch(self.reads.enc[:nr], self.reads.len[:nr])
        w, woff = pyoracle.fmi(oidx, sub, self.min_seed_len)
        assert np.array_equal(off[:nr + 1], woff) and all(np.array_equal(got[f], w[f]) for f in ("rid", "m", "n", "k", "l", "s")), \
            "fmi HIP output differs from the oracle"
        return f"bit-exact SMEM records vs oracle on the first {nr} reads ({k} SMEMs); offsets consistent on all {self.items}"

    def extra(self, ms_per_step):
        e = self.stats.get("ext_calls", 0)
        k = float(np.mean(self.kernel_ms))
        return {"backward_ext_per_step": e, "cp_occ_records_per_step": self.stats.get("cp_occ_records"),
                "smems_per_step": self.stats.get("smems"),
                "index_bytes": int(len(self.index.cp_occ)), "ref_mbp": self.ref_mbp,
                "g_ext_per_s": round(e / (k * 1e6), 3), "dominant_kernel": "fmi_seed_kernel", "dominant_kernel_ms": k}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        e = self.stats.get("ext_calls", 0)
        # readlen + 40 B x SMEMs streaming + the random index traffic.  8d prices an extension at two
        # 64-B CP_OCC records; GET_OCC reads ONE when both interval ends share a record (most extensions once the
        # interval is short), so the bytes counted here are 64 B x the records the kernel really fetched.
        recs = self.stats.get("cp_occ_records", 2 * e)
        alg = self.items * self.readlen + 40 * self.stats.get("smems", 0) + 64 * recs
        ach = alg / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "random 64-B CP_OCC records (counted by the kernel; extensions answered by the L2-resident short-pattern "
                        "table fetch none); measured chip ceiling for random 64-B reads is 55 G rec/s = 3.5 TB/s "
                        "(profiles/r01_random_read_ceiling.md)"}

    def host_roi(self, chunk=None):
        """the C driver's ROI (fmi/fmi.cpp:189-362 in the reference): read codes in, sorted SMEM records out, into page-locked
        arrays of the caller (gab_fmi_seed_into), one clone of the index handle per worker; like benchmarks/fmi/fmi.c a chunk is
        half a worker's share of the reads, 2^20 .. 2^22"""
        import ctypes as C
        from genarchbench_amd.fmi import SMEM_DTYPE
        n = self.items
        if chunk is None:
            chunk = min(max(-(-n // max(1, 2 * HOST_WORKERS)), 1 << 20), 1 << 22)
        d_out, d_off, total = self.result
        hip = C.CDLL("libamdhip64.so")
        off = np.zeros(n + 1, np.int64)
        assert hip.hipMemcpy(off.ctypes.data_as(C.c_void_p), C.c_void_p(d_off), C.c_size_t(8 * (n + 1)), C.c_int(2)) == 0
        nchunks = (n + chunk - 1) // chunk
        out = np.zeros(max(total, 1), SMEM_DTYPE)                    # chunk c writes at off[c * chunk]: the device path's layout
        enc, ln = self.reads.enc, self.reads.len
        pinned = pin(enc, ln, out)
        counts = [0] * nchunks

        def run(st, c):
            lo, hi = c * chunk, min(n, (c + 1) * chunk)
            counts[c] = st.seed_into(enc[lo:hi], ln[lo:hi], out[int(off[lo]):int(off[hi])], self.min_seed_len)
        try:
            sec = host_queue(nchunks, lambda: self.eng.clone(), run, lambda st: st.close())
        finally:
            unpin(pinned)
        assert sum(counts) == total, "host-pointer path and device path disagree on the SMEM count"
        dev = np.zeros(max(total, 1) * 40, np.uint8)
        assert hip.hipMemcpy(dev.ctypes.data_as(C.c_void_p), C.c_void_p(d_out), C.c_size_t(total * 40), C.c_int(2)) == 0
        dev = dev[:total * 40].view(SMEM_DTYPE)
        for c in range(nchunks):                                     # a chunk numbers its reads from 0 (fmi.cpp:340-343 adds the offset)
            lo, hi = int(off[c * chunk]), int(off[min(n, (c + 1) * chunk)])
            out["rid"][lo:hi] += np.uint32(c * chunk)
        assert all(np.array_equal(out[f][:total], dev[f]) for f in ("rid", "m", "n", "k", "l", "s")), "host-pointer path and device path disagree"
        return {"ms": round(sec * 1e3, 3), "value": round(n / sec / 1e6, 3), "unit": self.unit, "chunk": chunk, "workers_per_gpu": HOST_WORKERS,
                "note": "gab_fmi_seed_into on page-locked host arrays, chunks pulled by worker threads with clones of one index handle "
                        f"(the C driver's ROI): {enc.nbytes / 1e9:.2f} GB of read codes in, {total * 40 / 1e9:.2f} GB of SMEM records out; "
                        "every record equal to the device path's"}

    def cpu_baseline(self, cores):
        import ctypes as C
        from oracle import pyoracle
        from tools import gabgen
        exe = pyoracle.ref_path("fmi_ref")
        n = min(self.items, 400_000)
        sub = gabgen.ReadBatch(self.reads.enc[:n], self.reads.len[:n])
        if exe:
            with tempfile.TemporaryDirectory() as td:
                prefix = os.path.join(td, "ref")
                self.index.write(prefix, with_bns=True)
                fq = os.path.join(td, "reads.fq")
                gabgen.fmi_write_fastq(fq, sub)
                env = dict(os.environ, OMP_PROC_BIND="true", OMP_PLACES="cores")
                r = subprocess.run([exe, prefix, fq, "512", str(self.min_seed_len), str(cores)], capture_output=True,
                                   text=True, env=env)
                m = re.search(r"Computing time: ([\d.eE+-]+) s", r.stdout)
                if r.returncode == 0 and m and "realloc" not in r.stdout.split("totalSmems")[0]:
                    sec = float(m.group(1))
                    return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "reference",
                            "sample": f"first {n} reads of the same seeded input against the same index, reference fmi "
                                      f"<idx> <fq> 512 {self.min_seed_len} {cores}, its own Computing time ({sec:.2f} s)"}
                log("reference binary failed / hit its realloc path, using the oracle port:", r.stderr[-200:])
        oidx = pyoracle.FmIndex()
        cnt = (C.c_int64 * 5)(*[int(x) for x in self.index.count])
        pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(self.index.ref_seq_len), cnt,
                                              self.index.cp_occ.ctypes.data_as(C.c_void_p), C.c_int64(self.index.sentinel_index))
        t0 = time.time(); pyoracle.fmi(oidx, sub, self.min_seed_len, threads=cores); sec = time.time() - t0
        return {"value": round(n / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"first {n} reads, oracle/fmi.c + OpenMP ({sec:.2f} s)"}


class FmiWideWorkload(FmiWorkload):
    """the same workload with the interval-list format of indexes of 2^32 rows or more (a human genome: ~6.2 G rows) forced
    on the 256 Mbp index: 16-byte LDS list entries instead of 13, i.e. the occupancy the seeding kernel has at human scale"""
    name = "fmi-wide"
    metric = "fmi ROI M reads/sec (16-byte interval lists, the format of >= 2^32-row indexes)"
    wide_lists = True


_FMI_INDEX_CACHE = {}
_CTX = {}               # rank / world / dist / share_dir of this process (set in main)


class FmiSaWorkload(FmiWorkload):
    """the step after seeding: SMEM intervals -> reference coordinates through the compressed
    suffix array.  One unit = one coordinate; the SMEMs are those of `items` reads, seeded once outside the timed region
    and left on the device."""
    name = "fmi-sa"
    metric = "fmi SA look-up M coordinates/sec"
    unit = "M coordinates/s"
    default_items = 10_000_000
    max_occ = 500                                   # BWA-MEM2's default max_occ
    host_roi = None                                 # the reference driver has no ROI for this step (its call is commented out)

    def __init__(self, items, rank, dev, first=0, ids=None):
        super().__init__(items, rank, dev, first=first)
        self.eng.set_sa(self.index.sa_ms_byte, self.index.sa_ls_word)
        self.d_smems, _, self.nsmem = self.eng.seed_device(self.enc, self.len, self.min_seed_len)
        self.coords = 0
        self.sa_result = None

    def step(self, stream):
        self.sa_result = self.eng.get_sa_entries_device(self.d_smems, self.nsmem, self.max_occ, stream=stream)
        self.coords = self.sa_result[2]

    def units_per_step(self):
        return self.coords

    def after_step(self, timed):
        st = self.eng.last_sa_stats()
        if timed:
            self.kernel_ms.append(st["kernel_ms"])
        self.stats = st

    def check(self):
        import ctypes as C
        from oracle import pyoracle
        from genarchbench_amd.fmi import SMEM_DTYPE
        d_co, d_off, tot = self.sa_result
        hip = C.CDLL("libamdhip64.so")
        ns = min(200000, self.nsmem)
        sm = np.zeros(ns, SMEM_DTYPE); off = np.zeros(ns + 1, np.int64)
        assert hip.hipMemcpy(sm.ctypes.data_as(C.c_void_p), C.c_void_p(self.d_smems), C.c_size_t(40 * ns), C.c_int(2)) == 0
        assert hip.hipMemcpy(off.ctypes.data_as(C.c_void_p), C.c_void_p(d_off), C.c_size_t(8 * (ns + 1)), C.c_int(2)) == 0
        k = int(off[ns])
        got = np.zeros(max(k, 1), np.int64)
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(d_co), C.c_size_t(8 * k), C.c_int(2)) == 0
        oidx = pyoracle.FmIndex()
        cnt = (C.c_int64 * 5)(*[int(x) for x in self.index.count])
        pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(self.index.ref_seq_len), cnt,
                                              self.index.cp_occ.ctypes.data_as(C.c_void_p), C.c_int64(self.index.sentinel_index))
        pyoracle.lib().oracle_fmi_set_sa(C.byref(oidx), self.index.sa_ms_byte.ctypes.data_as(C.c_void_p),
                                         self.index.sa_ls_word.ctypes.data_as(C.c_void_p))
        want, woff, _ = pyoracle.fmi_sa_lookup(oidx, sm, self.max_occ)
        assert np.array_equal(off, woff) and np.array_equal(got[:k], want), "SA coordinates differ from the oracle"
        # property on a sample of all SMEMs: the coordinate really is an occurrence of the seed
        return f"bit-exact coordinates vs oracle on the first {ns} SMEMs ({k} coordinates) of {self.nsmem}"

    def extra(self, ms_per_step):
        k = float(np.mean(self.kernel_ms))
        return {"smems": self.nsmem, "coordinates_per_step": self.coords, "lf_steps_per_step": self.stats.get("lf_steps"),
                "max_occ": self.max_occ, "ref_mbp": self.ref_mbp, "dominant_kernel": "fmi_sa_kernel", "dominant_kernel_ms": k}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        # per coordinate: 40 B SMEM in (amortised), 8 B out, 5 B of sampled SA; per LF step one random 64-B CP_OCC record
        alg = 40 * self.nsmem + 13 * self.coords + 64 * self.stats.get("lf_steps", 0)
        ach = alg / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "one random 64-B CP_OCC record per LF step (mean 7 per coordinate) + two random reads of the sampled SA; "
                        "chip ceiling for random reads is ~55 G/s (profiles/r01_random_read_ceiling.md)"}

    def cpu_baseline(self, cores):
        import ctypes as C
        from oracle import pyoracle
        from genarchbench_amd.fmi import SMEM_DTYPE
        hip = C.CDLL("libamdhip64.so")
        ns = min(self.nsmem, 4_000_000)
        sm = np.zeros(ns, SMEM_DTYPE)
        assert hip.hipMemcpy(sm.ctypes.data_as(C.c_void_p), C.c_void_p(self.d_smems), C.c_size_t(40 * ns), C.c_int(2)) == 0
        oidx = pyoracle.FmIndex()
        cnt = (C.c_int64 * 5)(*[int(x) for x in self.index.count])
        pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(self.index.ref_seq_len), cnt,
                                              self.index.cp_occ.ctypes.data_as(C.c_void_p), C.c_int64(self.index.sentinel_index))
        pyoracle.lib().oracle_fmi_set_sa(C.byref(oidx), self.index.sa_ms_byte.ctypes.data_as(C.c_void_p),
                                         self.index.sa_ls_word.ctypes.data_as(C.c_void_p))
        os.environ["OMP_NUM_THREADS"] = str(cores)
        t0 = time.time(); coords, _, _ = pyoracle.fmi_sa_lookup(oidx, sm, self.max_occ); sec = time.time() - t0
        return {"value": round(len(coords) / sec / 1e6, 4), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"first {ns} SMEMs ({len(coords)} coordinates), oracle/fmi.c + OpenMP ({sec:.2f} s); the reference "
                          f"driver never calls get_sa_entries, so there is no reference binary to time"}


class ParseBswWorkload:
    """the bsw input text -> packed device buffers (what loadPairs does on the host, outside the
    ROI, main_banded.cpp:164-206).  One unit = one byte of input text; the text is resident in HBM."""
    name = "parse-bsw"
    metric = "bsw input parse GB/sec of text"
    unit = "GB/s"
    dtype = "u8"
    default_items = 10_000_000        # pairs
    seed = 2

    def __init__(self, items, rank, dev, first=0, ids=None):
        import torch
        from tools import gabgen
        from genarchbench_amd.parse import InputParser
        assert first == 0, "parse-bsw parses one file: run it at --gpus 1"
        self.items = items
        t0 = time.time()
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "pairs.txt")
            gabgen.write_text("bsw", p, self.seed, items, 0)
            self.text = np.fromfile(p, np.uint8)
        log(f"[rank {rank}] wrote and read back the bsw input text of {items} pairs ({len(self.text) / 1e9:.2f} GB) in {time.time() - t0:.1f}s")
        self.d_text = torch.from_numpy(self.text).to(dev)
        self.ps = InputParser(device=dev.index or 0)
        self.kernel_ms = []
        self.pk = None

    def step(self, stream):
        self.pk = self.ps.bsw_pairs_device(self.d_text.data_ptr(), self.d_text.numel(), stream=stream)

    def units_per_step(self):
        return len(self.text) / 1e3          # the harness divides by 1e6: GB/s = bytes / 1e9 / s

    def after_step(self, timed):
        if timed:
            self.kernel_ms.append(self.ps.last_stats()["kernel_ms"])

    def check(self):
        from tools import gabgen
        got = self.ps.bsw_to_host(self.pk)
        n = min(self.items, 200_000)
        want = gabgen.bsw(self.seed, n, 0)
        assert self.pk.n == self.items
        assert np.array_equal(got["len1"][:n], want.len1) and np.array_equal(got["len2"][:n], want.len2) and \
            np.array_equal(got["h0"][:n], want.h0), "parsed lengths / h0 differ from the generator's arrays"
        for i in range(0, n, 37):
            assert np.array_equal(got["ref"][got["ref_off"][i]:got["ref_off"][i] + got["len1"][i]],
                                  want.ref[want.ref_off[i]:want.ref_off[i] + want.len1[i]]), "parsed reference codes differ"
            assert np.array_equal(got["qry"][got["qry_off"][i]:got["qry_off"][i] + got["len2"][i]],
                                  want.qry[want.qry_off[i]:want.qry_off[i] + want.len2[i]]), "parsed query codes differ"
        assert (np.diff(got["ref_off"]) == ((got["len1"][:-1] + 3) & ~3)).all() and \
            (np.diff(got["qry_off"]) == ((got["len2"][:-1] + 3) & ~3)).all(), "offsets are not the padded prefix sums"
        return f"bit-exact lengths, h0 and codes vs the generator's arrays on the first {n} pairs (every 37th compared byte for byte); offsets consistent on all {self.items}"

    def out_bytes(self):
        return int(self.pk.ref_bytes + self.pk.qry_bytes + 28 * self.pk.n)

    def extra(self, ms_per_step):
        k = float(np.mean(self.kernel_ms))
        return {"pairs": self.items, "text_bytes": int(len(self.text)), "packed_bytes": self.out_bytes(),
                "m_pairs_per_s": round(self.items / (k * 1e3), 2), "dominant_kernel": "nl_fill / bsw_codes (5 streaming kernels)",
                "dominant_kernel_ms": k}

    def roofline(self):
        k = float(np.mean(self.kernel_ms))
        alg = len(self.text) + self.out_bytes()          # read the text once, write the packed buffers once
        ach = alg / (k * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": None,
                "note": "streaming: the text is actually read three times (newline count, newline fill, code copy) and the "
                        "8-byte line index (1 entry per ~70 B of text) written and read once; counted here: text once + outputs once"}

    def cpu_baseline(self, cores):
        """the C driver's line-by-line parser (same fgets / sscanf sequence as loadPairs) has no stand-alone entry point;
        time the equivalent single-threaded numpy-free Python-free path: the driver binary's own 'Read time' on a sample"""
        from oracle import pyoracle
        from tools import gabgen
        flags = open("/proc/cpuinfo").read()
        ref = pyoracle.ref_path("bsw_ref_" + ("avx512" if " avx512bw" in flags else "avx2"))
        exe, kind = (ref, "reference") if ref else (os.path.join(ROOT, "benchmarks", "bsw", "main_bsw"), "port")
        n = min(self.items, 1_000_000)
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "pairs.txt")
            gabgen.write_text("bsw", p, self.seed, n, 0)
            nbytes = os.path.getsize(p)
            r = subprocess.run([exe, "-pairs", p, "-t", str(cores), "-b", "512"], capture_output=True, text=True)
            m = re.search(r"Read time = ([\d.]+) s", r.stdout)
            if r.returncode == 0 and m and float(m.group(1)) > 0:
                sec = float(m.group(1))
                return {"value": round(nbytes / sec / 1e9, 4), "unit": self.unit, "cores": 1, "kind": kind,
                        "sample": f"{n} pairs ({nbytes / 1e6:.0f} MB): the 'Read time' line of "
                                  f"{'the compiled reference (loadPairs, main_banded.cpp:164-206)' if ref else 'this repo C driver'}"
                                  f" ({sec:.2f} s, single-threaded by construction)"}
        return {"value": None, "unit": self.unit, "cores": 1, "kind": "port", "sample": "driver binary not available"}


WORKLOADS = {"fmi-wide": FmiWideWorkload, "bitpal": BitpalWorkload, "bitpal-edit": BitpalEditWorkload, "parse-bsw": ParseBswWorkload, "fmi": FmiWorkload, "fmi-sa": FmiSaWorkload, "wfa": WfaWorkload, "bpm": BpmWorkload, "bsw": BswWorkload, "chain": ChainWorkload, "fast-chain": FastChainWorkload}

# The default run reports the whole metric of BASELINE.json ("M alignments/sec (bsw, bpm, wfa) + M seeds/sec (chain)"):
# the headline line is bsw-large (configs[1], the configuration the metric is quoted on) and `extra.suite` carries the other
# LARGE configurations (configs[2], configs[3], the fmi part of configs[4]) and the suite's SMALL inputs, each measured by
# the same code path with its own roofline, CPU baseline and parity verdict.  (name, workload, items or None = large, steps)
SUITE = [("chain-large", "chain", None, 5), ("fast-chain-large", "fast-chain", None, 5), ("bpm-large", "bpm", None, 5),
         ("wfa-large", "wfa", None, 5),
         ("bsw-small", "bsw", 100_000, 10), ("bpm-small", "bpm", 100_000, 10), ("wfa-small", "wfa", 100_000, 10),
         ("chain-small", "chain", 1000, 10), ("fast-chain-small", "fast-chain", 1000, 10),
         # the bpm driver's BitPAl modes, same 10 M pairs as bpm-large
         ("bitpal-edit-large", "bitpal-edit", None, 5), ("bitpal-scored-large", "bitpal", None, 3),
         ("fmi-large", "fmi", None, 3), ("fmi-large-wide-lists", "fmi-wide", None, 3)]
SUITE_BUDGET_S = float(os.environ.get("GAB_BENCH_BUDGET_S", "400"))    # entries that would start after this are skipped (and say so)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as a CHILD process
    (torch.distributed.run, one rank per GPU) before this process has made any GPU call, pass its output through and
    leave with its exit code.  A process that has touched the GPU is never replaced."""
    import torch                                   # importing torch / counting devices does not initialise the GPU
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    log("bench.py: starting", args.gpus, "ranks:", " ".join(cmd))
    raise SystemExit(subprocess.call(cmd, env=env))


def plan_shard(W, items, ctx, scaling):
    """-> (count, first, ids, total): what this rank processes.  `items` is the size of the FIXED input under strong scaling
    (split across the ranks) and the per-rank size under weak scaling."""
    from genarchbench_amd.shard import deal_longest_first, shard_range, shard_strong
    rank, world = ctx["rank"], ctx["world"]
    if ctx.get("emulate"):                                     # --shard R/N: rank R's share of an N-GPU strong-scaling run, on this one GPU
        rank, world = ctx["emulate"]
        scaling = "strong"
    if world == 1:
        return items, 0, None, items
    if scaling == "weak":
        first, count = shard_range(rank, world, items)
        return count, first, None, items * world
    if issubclass(W, ChainWorkload):
        # calls differ in size by three orders of magnitude: longest first, each to the rank with the least work so far
        # (what `omp for schedule(dynamic)` over the sorted calls does in effect, chain/src/host_kernel.cpp:98-105)
        from tools import gabgen
        ids = deal_longest_first(gabgen.chain_sizes(W.seed, items, 0, 50, 60000), world)[rank]
        return len(ids), 0, ids, items
    first, count = shard_strong(rank, world, items)
    return count, first, None, items


def run_workload(W, items, steps, warmup, ctx, args, with_cpu=True, with_host=True):
    """one workload, the bench contract's way: W untimed warm-up steps, K timed steps between barrier + synchronize on both
    sides, MAX over ranks.  Returns the result dict on rank 0 (None elsewhere)."""
    import torch
    rank, world, dev, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["dist"]
    scaling = args.scaling if world > 1 else "weak"            # at N = 1 the two coincide; the line says "weak" as before
    tw = time.time()
    mark = lambda what: log(f"[rank {rank}] {W.name}: {what} (+{time.time() - tw:.1f} s)")     # progress, one line per phase
    count, first, ids, total = plan_shard(W, items, ctx, scaling)
    wl = W(count, rank, dev, first=first, ids=ids)
    mark("inputs resident")
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        wl.step(stream)
        torch.cuda.synchronize()
        wl.after_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step(stream)
        wl.after_step(True)          # reads the step's HIP events (waits for the step, as the ROI does)
    barrier()
    elapsed = time.perf_counter() - t0
    from genarchbench_amd.shard import aggregate
    units = wl.units_per_step() if hasattr(wl, "units_per_step") else getattr(wl, "items", count)
    own_ms = elapsed / steps * 1e3
    elapsed, total_units = aggregate(elapsed, units, dist if world > 1 else None, ctx.get("agg_dev", dev))

    mark(f"{steps} timed steps done")
    verdict = None if args.no_check else wl.check()
    mark("parity check done")
    per_rank = None
    if world > 1:                                             # who was the slowest, and on how much work (informative)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "units": float(units), "ms_per_step": round(own_ms, 4),
                                          "kernel_ms": round(float(np.mean(wl.kernel_ms)), 4) if getattr(wl, "kernel_ms", None) else None})
    out = None
    if rank == 0:
        ms = elapsed / steps * 1e3
        value = total_units / (ms * 1e-3) / 1e6       # units of all ranks / max-over-ranks time
        large = items == W.default_items
        if ctx.get("emulate"):
            out_shard = "rank %d's share of a %d-GPU strong-scaling run (--shard), processed on ONE GPU: %d of %d items" % (
                ctx["emulate"][0], ctx["emulate"][1], count if ids is None else len(ids), items)
        sharding = (out_shard if ctx.get("emulate") else "1 GPU" if world == 1 else
                    f"strong: the fixed input of {total} items split across {world} ranks" +
                    (", calls dealt longest first to the least-loaded rank" if ids is not None else ", contiguous id ranges") +
                    ", no collective" if scaling == "strong" else
                    f"weak: {world} x independent id ranges of {items} items, no collective")
        if ctx.get("share"):
            sharding += " -- TEST MODE: all ranks share GPU 0 (GAB_BENCH_SHARE_GPU=1), not an N-GPU measurement"
        out = {
            "metric": W.metric, "value": round(value, 4), "unit": W.unit, "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": W.dtype,
            "data": "synthetic (seeded generator tools/gen, SURVEY.md 8d distributions)",
            "config": {"workload": f"{W.name}-large" if large else f"{W.name}-{items}",
                       "total_items": total, "items_per_gpu": round(total / world, 1), "sharding": sharding,
                       # (the task's measurement rule: `value` = throughput with the inputs resident in HBM; the PCIe-inclusive
                       # rate is reported beside it and is what the comparison with the CPU baseline uses)
                       "value_is": "hbm_resident: inputs already in HBM when the timed region starts.  The drop-in ROI of SURVEY.md 8(d) -- "
                                   "host pointers in and out, PCIe both ways -- is extra.value_roi_incl_pcie, and extra.x_cpu_baseline compares "
                                   "THAT with the CPU baseline (whose ROI includes its marshalling); extra.x_cpu_baseline_hbm_resident is the kernel-only ratio"},
            "roofline": wl.roofline(), "extra": wl.extra(ms), "parity": verdict,
        }
        out["extra"]["value_hbm_resident"] = out["value"]
        if per_rank:
            out["extra"]["per_rank"] = per_rank
        km = out["extra"].get("dominant_kernel_ms")
        t, detail = pmc_traffic(W.name, large and world == 1 and not ctx.get("emulate"), km)
        if t is not None:
            out["roofline"]["traffic"] = t                  # GB/s of real HBM traffic, comparable with `achieved`
            out["roofline"]["traffic_detail"] = detail
        # BASELINE.md 3.5 also asks for the fraction of the MEASURED copy bandwidth (6.29 TB/s, MI355X_MICROARCH.md)
        if out["roofline"].get("unit") == "GB/s" and out["roofline"].get("achieved") is not None:
            out["roofline"]["frac_of_measured_copy_6290"] = round(out["roofline"]["achieved"] / 6290.0, 6)
        if with_host and world == 1 and callable(getattr(wl, "host_roi", None)) and not args.no_host_roi:
            # SURVEY.md 8d's ROI -- what a drop-in driver times: host pointers in, host pointers out (PCIe both ways); never `value`
            try:
                out["extra"]["roi_incl_pcie"] = wl.host_roi()
                out["extra"]["value_roi_incl_pcie"] = out["extra"]["roi_incl_pcie"].get("value")
            except Exception as e:      # the figure is informative; a failure must not lose the measured line
                out["extra"]["roi_incl_pcie"] = {"error": str(e)[:300]}
            mark("host-pointer ROI done")
        if world > 1:
            out["cpu_baseline"] = None                     # timed at N = 1 only (the other ranks would wait for rank 0's host cores)
        if with_cpu and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = wl.cpu_baseline(host_cores())
            out["cpu_baseline"]["host_threads_visible"] = os.cpu_count()      # `cores` of these were used (cgroup quota)
            mark("cpu baseline done")
            finish_cpu_ratios(out, world, os.cpu_count())
    del wl
    torch.cuda.empty_cache()
    return out


def finish_cpu_ratios(out, world, host_threads, quota=None):
    """per-core figure, the quota in words, and the ratios GPU / CPU: like for like -- the CPU figure is the reference's
    ROI, marshalling included, so the GPU figure beside it is the drop-in ROI including PCIe; the HBM-resident ratio is kept under
    its own name"""
    cb, v = out["cpu_baseline"], out["value"]
    if not cb or not cb.get("value"):
        return
    cores = max(int(cb.get("cores") or 1), 1)
    cb["per_core"] = round(cb["value"] / cores, 4)
    quota = quota or host_cores()
    if quota < (host_threads or quota):
        used = "" if cores == quota else f", {cores} thread(s) of it used"
        cb["sample"] = f"{quota}-core cgroup quota of a {host_threads}-thread host{used}; " + str(cb.get("sample", ""))
    out["extra"]["x_cpu_baseline_hbm_resident"] = round(v / world / cb["value"], 2)
    if out["extra"].get("value_roi_incl_pcie"):
        out["extra"]["x_cpu_baseline"] = round(out["extra"]["value_roi_incl_pcie"] / cb["value"], 2)
        out["extra"]["x_cpu_baseline_is"] = "value_roi_incl_pcie / cpu_baseline.value (one GPU incl. PCIe vs the reference on the host's cores)"
        out["extra"]["cpu_cores_equivalent"] = round(out["extra"]["value_roi_incl_pcie"] / cb["per_core"], 1)
    else:
        out["extra"]["x_cpu_baseline"] = None      # no host-pointer ROI measured in this run: no like-for-like figure


# ---- output: one compact headline as the LAST line of stdout, one line per suite entry before it ------------------------
HEADLINE_LIMIT = 4096


def _cut(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact(r, note_chars=160):
    """the fields of a result the driver's record needs, bounded in size (the full dict goes to bench_suite.json)"""
    keep = {k: r.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype") if k in r}
    keep["data"] = "synthetic"
    cfg = r.get("config") or {}
    keep["config"] = {"workload": cfg.get("workload"), "total_items": cfg.get("total_items"), "items_per_gpu": cfg.get("items_per_gpu"),
                      "sharding": _cut(cfg.get("sharding", ""), 150), "value_is": "hbm_resident"}
    rf = r.get("roofline") or {}
    keep["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    cb = r.get("cpu_baseline")
    keep["cpu_baseline"] = None if not cb else {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                                "per_core": cb.get("per_core"), "host_threads_visible": cb.get("host_threads_visible"), "kind": cb.get("kind"),
                                                "sample": _cut(cb.get("sample", ""), note_chars)}
    keep["parity"] = _cut(r.get("parity"), 120) if r.get("parity") is not None else None
    ex = r.get("extra") or {}
    keep["extra"] = {k: ex[k] for k in ("dominant_kernel", "dominant_kernel_ms", "value_hbm_resident", "value_roi_incl_pcie",
                                        "x_cpu_baseline", "x_cpu_baseline_hbm_resident", "cpu_cores_equivalent", "gcups") if ex.get(k) is not None}
    if "x_cpu_baseline" in keep["extra"]:
        keep["extra"]["x_cpu_baseline_is"] = "roi_incl_pcie / cpu_baseline"
    if isinstance(keep["extra"].get("dominant_kernel"), str):
        keep["extra"]["dominant_kernel"] = _cut(keep["extra"]["dominant_kernel"], 60)
    if isinstance(keep["extra"].get("dominant_kernel_ms"), float):
        keep["extra"]["dominant_kernel_ms"] = round(keep["extra"]["dominant_kernel_ms"], 4)
    return keep


def headline(out, suite):
    """the LAST stdout line: the compact headline + one [value, unit, roofline frac, value incl. PCIe, x CPU] row per suite entry"""
    h = compact(out)
    if suite:
        rows = {}
        for name, r in suite.items():
            if "value" in r:
                ex = r.get("extra") or {}
                rows[name] = [r["value"], r.get("unit"), (r.get("roofline") or {}).get("frac"), ex.get("value_roi_incl_pcie"),
                              ex.get("x_cpu_baseline")]
            else:
                rows[name] = _cut(r.get("skipped") or r.get("error") or "?", 60)
        h["extra"]["suite"] = rows
        h["extra"]["suite_columns"] = ["value_hbm_resident", "unit", "roofline.frac", "value_roi_incl_pcie", "x_cpu_baseline (roi_incl_pcie / cpu)"]
        h["extra"]["suite_detail"] = "bench_suite.json + the {\"suite\": ...} lines above"
    line = json.dumps(h, separators=(",", ":"))
    if len(line) >= HEADLINE_LIMIT:                      # never again an unparseable record: shed the optional parts
        for k in ("suite_detail", "suite_columns", "suite"):
            h["extra"].pop(k, None)
            line = json.dumps(h, separators=(",", ":"))
            if len(line) < HEADLINE_LIMIT:
                break
    assert len(line) < HEADLINE_LIMIT, "headline does not fit"
    return line


def emit(out, suite, path=None):
    """suite lines first, then the headline as the very last thing on stdout; the full detail to bench_suite.json"""
    for name, r in (suite or {}).items():
        row = {"suite": name}
        row.update(compact(r) if "value" in r else r)
        print(json.dumps(row, separators=(",", ":")), flush=True)
    path = path or os.environ.get("GAB_BENCH_SUITE_JSON") or os.path.join(ROOT, "bench_suite.json")
    try:
        with open(path, "w") as f:
            json.dump({"headline": out, "suite": suite or {}}, f, indent=1)
    except OSError as e:
        log(f"could not write {path}: {e}")
    sys.stdout.flush()
    print(headline(out, suite), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="one workload only (default: bsw-large as the headline + the suite, one line each)")
    ap.add_argument("--items", type=int, default=0,
                    help="size of the input (default: the large config): the whole input under strong scaling, per GPU under weak scaling")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="with --gpus N > 1: strong = ONE large input split across the N GPUs (default, BASELINE.json configs[4]); "
                         "weak = every GPU gets a full-size input of its own")
    ap.add_argument("--shard", default=None, metavar="R/N",
                    help="(one GPU) process only rank R's share of an N-GPU strong-scaling run: what that rank would spend; "
                         "T / max over R of these times is the N-GPU figure the sharding can reach")
    ap.add_argument("--no-suite", action="store_true", help="headline only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-roi", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    # the GPU box shows every hardware thread of the host but grants a CPU quota: without this the OpenMP generators and
    # the oracle would start one thread per visible CPU (256 threads on a 16-core quota)
    os.environ.setdefault("OMP_NUM_THREADS", str(host_cores()))

    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        self_launch(args)                          # does not return
    world = max(world, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` "
                         f"or under torch.distributed.run with --nproc-per-node equal to --gpus")

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libgab_hip has no CPU fallback")
    if world > 1:                                  # the host's cores are shared by the ranks (generators, oracle checks)
        os.environ["OMP_NUM_THREADS"] = str(max(1, host_cores() // world))
    # GAB_BENCH_SHARE_GPU=1 (tests only): the N ranks all use GPU 0 and rendezvous over gloo -- the N > 1 code path (sharding,
    # barrier, MAX / SUM aggregation) with real engine handles on a one-GPU box.  The line says so.
    share = world > 1 and os.environ.get("GAB_BENCH_SHARE_GPU") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 and not share else 0)
    torch.cuda.set_device(dev)
    ctx = {"rank": rank, "world": world, "dev": dev, "dist": dist, "agg_dev": None if share else dev, "share": share}
    if args.shard:
        if world != 1:
            raise SystemExit("bench.py: --shard emulates one rank of an N-GPU run on ONE GPU (use it with --gpus 1)")
        r_, n_ = (int(v) for v in args.shard.split("/"))
        assert 0 <= r_ < n_
        ctx["emulate"] = (r_, n_)
    share_dir = None
    if world > 1:                                  # a directory all ranks of this node see (rank 0's FM-index file)
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        share_dir = os.path.join(base, f"gab_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}")
        os.makedirs(share_dir, exist_ok=True)
    _CTX.update(rank=rank, world=world, dist=dist, share_dir=share_dir)
    t_start = time.time()

    try:
        W = WORKLOADS[args.workload or "bsw"]
        out = run_workload(W, args.items or W.default_items, args.steps, args.warmup, ctx, args)

        suite = {}
        if args.workload is None and not args.items and not args.no_suite and not args.shard:
            for name, wname, items, steps in SUITE:
                go = [time.time() - t_start < SUITE_BUDGET_S]
                if world > 1:
                    dist.broadcast_object_list(go, src=0)           # every rank takes the same branch
                if not go[0]:
                    suite[name] = {"skipped": f"time budget of {SUITE_BUDGET_S:.0f} s reached (GAB_BENCH_BUDGET_S); run --workload {wname}"}
                    continue
                if world > 1 and items is not None:
                    suite[name] = {"skipped": "small inputs run at --gpus 1 (they do not fill one GPU)"}
                    continue
                SW = WORKLOADS[wname]
                t0 = time.time()
                try:
                    r = run_workload(SW, items or SW.default_items, min(steps, max(args.steps, 1)), 1, ctx, args,
                                     with_host=items is None)
                except Exception as e:
                    if world > 1:
                        raise
                    log(f"suite entry {name} failed: {e}")
                    suite[name] = {"error": str(e)[:300]}
                    continue
                if r is not None:
                    r["wall_s"] = round(time.time() - t0, 1)
                    suite[name] = r
                    log(f"suite: {name}: {r['value']} {r['unit']} ({r['wall_s']} s)")
        if rank == 0:
            emit(out, suite)
    finally:
        if world > 1:
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:       # noqa: BLE001 -- a failed rank must not hide the first error behind a barrier time-out
                pass
            if rank == 0 and share_dir:
                import shutil
                shutil.rmtree(share_dir, ignore_errors=True)


if __name__ == "__main__":
    main()
