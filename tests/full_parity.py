#!/usr/bin/env python3
"""Full-size parity: every item of the LARGE bench configurations against the oracle (run on the GPU box, repo root).

bench.py checks a slice per run (its check is outside the timed region but still has to stay short); this is the one-off
exhaustive version: all 10 M bsw / bpm / bitpal pairs, all 1 M wfa pairs (scores, lengths and every CIGAR byte), all
10 000 chain / fast-chain calls, all 10 M fmi reads (77 M SMEM records).  Writes a summary to stdout;
profiles/rNN_full_size_parity.md keeps the last one of each round.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle      # noqa: E402  (the checker; this script is test infrastructure)
from tools import gabgen         # noqa: E402


def line(name, n, what, ok, t_gpu, t_cpu):
    print(f"| {name} | {n} | {what} | {'identical' if ok else 'MISMATCH'} | {t_gpu:.2f} s | {t_cpu:.1f} s |", flush=True)
    return ok


def main():
    which = sys.argv[1:] or ["bsw", "bpm", "bitpal", "wfa", "chain", "fast-chain", "fmi"]
    ok = True
    print("| workload | items | compared | result | GPU (host-pointer entry point, incl. PCIe) | oracle, all host threads |\n|---|---|---|---|---|---|")
    if "bsw" in which:
        from genarchbench_amd.bsw import BandedPairWiseSW
        b = gabgen.bsw(2, 10_000_000, 0)
        e = BandedPairWiseSW()
        t0 = time.time(); got = e.getScores16(b); t1 = time.time()
        want = pyoracle.bsw(b)[:, 0]; t2 = time.time()
        ok &= line("bsw-large", b.n, "score of every pair", np.array_equal(got, want), t1 - t0, t2 - t1)
        e.close(); del b, got, want
    if "bpm" in which or "bitpal" in which:
        raw = gabgen.pairs(3, 10_000_000, 0, 151)
        b = raw.swapped_combined()
        if "bpm" in which:
            from genarchbench_amd.bpm import BpmEngine
            e = BpmEngine()
            t0 = time.time(); got = e.benchmark_edit_bpm(b); t1 = time.time()
            want = pyoracle.bpm(b); t2 = time.time()
            ok &= line("bpm-large", b.n, "score of every pair", np.array_equal(got, want), t1 - t0, t2 - t1)
            e.close()
        if "bitpal" in which:
            from genarchbench_amd.bitpal import BitpalEngine
            for alg, nm in ((1, "bitpal-large (scored)"), (0, "bitpal-edit-large")):
                e = BitpalEngine(alg)
                t0 = time.time(); got = e.benchmark_bitpal(b); t1 = time.time()
                want = pyoracle.bitpal(b, alg); t2 = time.time()
                ok &= line(nm, b.n, "score of every pair", np.array_equal(got, want), t1 - t0, t2 - t1)
                e.close()
        del raw, b
    if "wfa" in which:
        from genarchbench_amd.wfa import AffineWavefronts
        b = gabgen.pairs(4, 1_000_000, 0, 151)
        for red, nm in ((None, "wfa-large"), ((10, 50), "wfa-large, adaptive (10, 50)")):
            e = AffineWavefronts() if red is None else AffineWavefronts(min_wavefront_length=red[0], max_distance_threshold=red[1])
            t0 = time.time(); go, goff, gl, gs = e.align(b); t1 = time.time()
            wo, woff, wl, ws = pyoracle.wfa(b, reduction=red); t2 = time.time()
            same = np.array_equal(gs, ws) and np.array_equal(gl, wl) and np.array_equal(goff, woff)
            if same:
                mask = np.zeros(len(wo), bool)
                idx = np.repeat(woff, wl) + (np.arange(int(wl.sum())) - np.repeat(np.cumsum(wl) - wl, wl))
                mask[idx] = True
                same = np.array_equal(go[:len(wo)][mask], wo[mask])
            ok &= line(nm, b.n, "score, length and every CIGAR operation", same, t1 - t0, t2 - t1)
            e.close()
        del b
    for mode, nm in ((0, "chain"), (1, "fast-chain")):
        if nm not in which:
            continue
        from genarchbench_amd.chain import ChainEngine
        cb = gabgen.chain(5, 10000, 0)
        e = ChainEngine(device=0)
        t0 = time.time(); s, p = e.host_chain_kernel(cb, mode); t1 = time.time()
        ws, wp = pyoracle.chain(cb, mode); t2 = time.time()
        ok &= line(f"{nm}-large", f"{len(cb.hdr)} calls, {cb.nanchors} anchors", "score and parent of every anchor",
                   np.array_equal(s, ws) and np.array_equal(p, wp), t1 - t0, t2 - t1)
        e.close(); del cb
    for mode, nm in ((0, "chain"), (1, "fast-chain")):
        if nm + "-shard" not in which:
            continue
        # one rank's share of chain-large under strong scaling on 8 GPUs (calls dealt longest first): a batch that waits for its
        # longest call, so gab_chain_run_device hands most of it to the latency form (chain_fast_kernel) -- every anchor compared
        from genarchbench_amd.chain import ChainEngine
        from genarchbench_amd.shard import deal_longest_first
        ids = deal_longest_first(gabgen.chain_sizes(5, 10000, 0, 50, 60000), 8)[0]
        cb = gabgen.chain_ids(5, ids, 0, 50, 60000)
        os.environ["GAB_CHAIN_NO_OVERLAP"] = "1"          # plain copies + gab_chain_run_device: the path bench.py's ranks take
        try:
            e = ChainEngine(device=0)
            t0 = time.time(); s, p = e.host_chain_kernel(cb, mode); t1 = time.time()
        finally:
            os.environ.pop("GAB_CHAIN_NO_OVERLAP", None)
        ws, wp = pyoracle.chain(cb, mode); t2 = time.time()
        ok &= line(f"{nm}-large, shard 0 of 8 (latency form)", f"{len(cb.hdr)} calls, {cb.nanchors} anchors", "score and parent of every anchor",
                   np.array_equal(s, ws) and np.array_equal(p, wp), t1 - t0, t2 - t1)
        e.close(); del cb
    if "fmi-mid" in which:
        # fmi at a size that fits the driver-run test suite: 48 Mbp reference (96 M BWT rows, 92 MiB of CP_OCC --
        # larger than the 32 MiB of L2; parity does not need the index to leave the Infinity Cache), 1.5 M reads of 151 bp,
        # every field of every SMEM, with the 13-byte interval lists and with the 16-byte ones of >= 2^32-row indexes
        import ctypes as C
        from tools import mkindex
        from genarchbench_amd.fmi import FMI_search
        ref = gabgen.fmi_ref(16, 48_000_000, 5)
        idx = mkindex.FmIndex(ref)
        reads = gabgen.fmi_reads(17, ref, 1_500_000, 151, 151)
        oidx = pyoracle.FmIndex()
        cnt = (C.c_int64 * 5)(*[int(x) for x in idx.count])
        pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(idx.ref_seq_len), cnt, idx.cp_occ.ctypes.data_as(C.c_void_p),
                                              C.c_int64(idx.sentinel_index))
        t1 = time.time(); w, woff = pyoracle.fmi(oidx, reads, 19); t2 = time.time()
        for wide, nm in ((False, "fmi-large at 48 Mbp, 13-byte lists"), (True, "fmi-large at 48 Mbp, 16-byte lists")):
            if wide:
                os.environ["GAB_FMI_WIDE_LISTS"] = "1"
            try:
                e = FMI_search(arrays=(idx.ref_seq_len, idx.count, idx.cp_occ, idx.sentinel_index))
            finally:
                os.environ.pop("GAB_FMI_WIDE_LISTS", None)
            t0 = time.time(); sm, off = e.seed(reads, 19); tg = time.time() - t0
            same = np.array_equal(off, woff) and len(sm) == len(w) and all(np.array_equal(sm[f], w[f]) for f in ("rid", "m", "n", "k", "l", "s"))
            ok &= line(nm, f"{reads.n} reads, {len(w)} SMEMs", "all six fields of every SMEM, per-read offsets", same, tg, t2 - t1)
            e.close()
        del ref, idx, reads, w
    if "fmi" in which:
        # fmi-large: all 10 M reads against the 256 Mbp index (bench.py checks the first 20 000 per run)
        import ctypes as C
        from tools import mkindex
        from genarchbench_amd.fmi import FMI_search
        ref = gabgen.fmi_ref(6, 256_000_000, 5)
        idx = mkindex.FmIndex(ref)
        reads = gabgen.fmi_reads(7, ref, 10_000_000, 151, 151)
        e = FMI_search(arrays=(idx.ref_seq_len, idx.count, idx.cp_occ, idx.sentinel_index))
        t0 = time.time(); sm, off = e.seed(reads, 19); t1 = time.time()
        oidx = pyoracle.FmIndex()
        cnt = (C.c_int64 * 5)(*[int(x) for x in idx.count])
        pyoracle.lib().oracle_fmi_from_arrays(C.byref(oidx), C.c_int64(idx.ref_seq_len), cnt, idx.cp_occ.ctypes.data_as(C.c_void_p),
                                              C.c_int64(idx.sentinel_index))
        w, woff = pyoracle.fmi(oidx, reads, 19); t2 = time.time()
        same = np.array_equal(off, woff) and len(sm) == len(w) and all(np.array_equal(sm[f], w[f]) for f in ("rid", "m", "n", "k", "l", "s"))
        ok &= line("fmi-large (256 Mbp index)", f"{reads.n} reads, {len(w)} SMEMs", "all six fields of every SMEM, per-read offsets", same, t1 - t0, t2 - t1)
        e.close()
    print("ALL IDENTICAL" if ok else "MISMATCH FOUND")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
