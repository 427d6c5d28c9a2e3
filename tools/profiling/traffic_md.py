"""profiles/<round>_hbm_traffic.json (+ the previous round's) -> the rows of profiles/<round>_hbm_traffic.md on stdout:
    python tools/profiling/traffic_md.py r04 r03"""
import json, os, sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rnd, prev = sys.argv[1], sys.argv[2]
a = json.load(open(os.path.join(root, "profiles", f"{rnd}_hbm_traffic.json")))
b = json.load(open(os.path.join(root, "profiles", f"{prev}_hbm_traffic.json")))
print(f"| workload | kernel | launches | FETCH_SIZE raw {rnd} (GB) | WRITE_SIZE {rnd} (GB) | {prev} (fetch / write) |")
print("|---|---|---|---|---|---|")
for w in sorted(a):
    for k, v in a[w].items():
        if v["fetch_gb_raw"] + v["write_gb"] < 0.01:
            continue
        o = b.get(w, {}).get(k)
        print(f"| {w} | `{k}` | {v['launches']} | {v['fetch_gb_raw']:.2f} | {v['write_gb']:.2f} | " + (f"{o['fetch_gb_raw']:.2f} / {o['write_gb']:.2f}" if o else "—") + " |")
