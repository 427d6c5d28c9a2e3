#!/usr/bin/env python3
"""Timeline of the kernels of the LAST step in a rocprofv3 kernel trace (csv): start / end in ms relative to the step's first kernel.
    python tools/profiling/kernel_timeline.py <kernel_trace.csv> [first-kernel-name-substring]
The step is found as the last occurrence of the kernel whose name contains the substring (default: chain_facts_kernel)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "chain_facts_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    print(f"{a:8.3f} {b:8.3f}  {b - a:7.3f} ms  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d}  {name}")
