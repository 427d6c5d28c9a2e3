"""Sum rocprofv3 --pmc counter_collection.csv per kernel: python pmc_sum.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k: continue
        acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k[:60], r["Counter_Name"])] += 1
for k, v in acc.items():
    print(k)
    for cn, val in sorted(v.items()):
        print(f"   {cn:32s} {val:.6g}  (launches {n[(k, cn)]})")
