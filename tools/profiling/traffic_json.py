"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collections of tools/profiling/hbm_traffic.sh -> the JSON table bench.py reads
(profiles/rNN_hbm_traffic.json): {workload: {kernel: {launches, fetch_gb_raw, write_gb}}}.  Counters are in KiB
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section); FETCH_SIZE is stored raw -- bench.py applies the x2 of wide coalesced
streams where it is calibrated.      python traffic_json.py <dir with <workload>_FETCH_SIZE/ and <workload>_WRITE_SIZE/>"""
import collections, csv, glob, json, os, re, sys

d = sys.argv[1]
out = {}
for path in sorted(glob.glob(os.path.join(d, "*_FETCH_SIZE"))):
    w = os.path.basename(path)[:-len("_FETCH_SIZE")]
    tab = collections.defaultdict(lambda: {"launches": 0, "fetch_gb_raw": 0.0, "write_gb": 0.0})
    for counter, key in (("FETCH_SIZE", "fetch_gb_raw"), ("WRITE_SIZE", "write_gb")):
        # gpurun merges every call's files into the same directory: only the NEWEST collection counts (two would be summed)
        for f in sorted(glob.glob(os.path.join(d, f"{w}_{counter}", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                k = re.sub(r"\(.*$", "", k).strip()                     # drop the argument list, keep template arguments
                tab[k][key] += float(r["Counter_Value"]) * 1024 / 1e9
                if counter == "FETCH_SIZE":
                    tab[k]["launches"] += 1
    out[w] = {k: {"launches": v["launches"], "fetch_gb_raw": round(v["fetch_gb_raw"], 4), "write_gb": round(v["write_gb"], 4)} for k, v in tab.items()}
json.dump(out, sys.stdout, indent=1)
