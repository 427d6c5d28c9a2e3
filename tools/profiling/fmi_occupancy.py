#!/usr/bin/env python3
"""fmi_seed_kernel: throughput against the occupancy the LDS interval lists allow ("the lists in LDS cap
occupancy").  One 256 Mbp index and 10 M reads, built once; the seeding kernel with 6 .. 16 list entries per lane in LDS
($GAB_FMI_LDS_ENTRIES, read when the handle is made; entries that do not fit spill to the lane's global scratch) and with
the 13- / 16-byte entry formats.  Prints a markdown table (profiles/r03_fmi_occupancy.md).
    python tools/profiling/fmi_occupancy.py [ref_mbp] [reads]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np      # noqa: E402
import torch            # noqa: E402
from tools import gabgen, mkindex                     # noqa: E402
from genarchbench_amd.fmi import FMI_search           # noqa: E402

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda", 0)
ref = gabgen.fmi_ref(6, mbp * 1_000_000, 5)
idx = mkindex.FmIndex(ref)
reads = gabgen.fmi_reads(7, ref, nreads, 151, 151)
enc = torch.from_numpy(reads.enc).to(dev); ln = torch.from_numpy(reads.len).to(dev)
print(f"| list entries in LDS | entry bytes | LDS per wave | waves per CU the LDS allows | kernel ms | M reads/s | G CP_OCC records/s |\n|---|---|---|---|---|---|---|", flush=True)
for wide in (0, 1):
    for entries in (4, 6, 8, 10, 12, 14, 16):
        os.environ["GAB_FMI_LDS_ENTRIES"] = str(entries)
        if wide:
            os.environ["GAB_FMI_WIDE_LISTS"] = "1"
        else:
            os.environ.pop("GAB_FMI_WIDE_LISTS", None)
        e = FMI_search(arrays=(idx.ref_seq_len, idx.count, idx.cp_occ, idx.sentinel_index))
        ms = []
        for _ in range(3):
            e.seed_device(enc, ln, 19)
            ms.append(e.last_stats()["kernel_ms"])
        st = e.last_stats()
        k = min(ms[1:])
        eb = 16 if wide else 13
        lds = ((151 + 7) // 8 * 64 * 4 + 15) // 16 * 16 + (entries * 64 * 16 if wide else (entries * 64 * 13 + 15) // 16 * 16)
        print(f"| {entries} | {eb} | {lds} B | {min(32, 160 * 1024 // lds)} | {k:.1f} | {nreads / k / 1e3:.2f} | {st['cp_occ_records'] / k / 1e6:.1f} |", flush=True)
        e.close()
