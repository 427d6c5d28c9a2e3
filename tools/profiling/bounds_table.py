"""VALU issue / lane utilisation / occupancy / wait share per kernel from the two PMC passes of tools/profiling/valu_util.sh.

    python bounds_table.py <gpurun_out/valu> <workload> [<kernel ms from the un-profiled bench line>]

VALU busy = SQ_INSTS_VALU x 4 clk / (1024 SIMDs x kernel time x 2.4 GHz)   (4 clk: the issue cost of the slow instruction
class; kernels made of 2-clk instructions can exceed what this suggests -- see profiles/r02_valu_issue.md);
lanes = SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU x 64) ... reported by the counter in units of 4 lane-cycles per thread-cycle;
waves/CU = SQ_WAVE_CYCLES x 4 / kernel clk / 256; waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.  Kernel time: the duration of the
kernel in the PMC run's own kernel trace (profiled runs are a few per cent slower than plain ones)."""
import collections, csv, glob, os, re, sys

d, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
for p in (0, 1):
    # (gpurun merges every call's files into the same directory: the NEWEST collection only)
    for f in sorted(glob.glob(os.path.join(d, f"{w}_p{p}", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if p == 0:
        for f in sorted(glob.glob(os.path.join(d, f"{w}_p{p}", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
                dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
print("| workload | kernel | ms (sum of launches, profiled run) | VALU busy | lanes | waves/CU | waiting | SALU/VALU |\n|---|---|---|---|---|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -dur.get(kv[0], 0)):
    if k.startswith("__amd") or k.startswith("at::") or not v.get("SQ_INSTS_VALU") or dur.get(k, 0) < 0.05:
        continue
    ms = dur[k]
    clk = ms * 1e-3 * 2.4e9
    busy = v["SQ_INSTS_VALU"] * 4 / (1024 * clk)
    lanes = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_INSTS_VALU"] * 64)
    waves = v["SQ_WAVE_CYCLES"] * 4 / clk / 256
    wait = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    print(f"| {w} | `{k}` | {ms:.2f} | {100 * busy:.0f} % | {100 * lanes:.0f} % | {waves:.1f} | {100 * wait:.0f} % | {v['SQ_INSTS_SALU'] / v['SQ_INSTS_VALU']:.2f} |")
